import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rkfd_pkg; R = rkfd_pkg.load()
import numpy as np
warm = int(os.environ.get('WARM','20'))
for cfg in sys.argv[1:]:
    sc = R.scenarios.CONFIGS[cfg](batch=int(os.environ.get('BATCH','4096')))
    b = R.Batch(sc['world'], int(os.environ.get('BATCH','4096')), max_rigid=sc['max_rigid'])
    b.set_state(sc['dis'], sc['vel']); b.update_init(); b.update(warm)
    p = b.profile(5).astype(np.float64)/5
    names=['kin','cd+pen','sweep2','sweep3','mlcp','tail','m:entries','launch total','s2:pre','s2:gather','s2:UD','s2:rank1','s2:store','s2:chol','m:tgt+b','m:probe-up','k:local','k:fkscan','k:axis','k:velscan','k:bias+friction','m:pgs','m:setforce','m:delta-in','q:setup','q:rows+W','q:S','q:cholS+solve','q:x','q:mult+drop','q:step+cycle','m:group-layout']
    if sc['world'].model.contents.solver == 2: names[6]='v:matrix'; names[14]='v:b+tgt'; names[15]='v:probes'; names[23]='v:delta-in'; names[24:32]=['v:qp-create','v:qp-solve','v:setforce+nfc','v:friction-lp','c:vertex-test','c:clip+centre','c:integrals','c:merge+vel']
    act = b.get_contact()[0]; print(cfg, 'warm', warm, 'mean contacts', act.sum(1).mean(), 'lds', b.lds_bytes, ' '.join('%s=%.0f'%(n,x) for n,x in zip(names,p.mean(0))), 'sum', p[:,:6].sum(1).mean())
