"""arm_wall scenario step by step: GPU batch (one instance, host controller) against the oracle; prints the deviation per step"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rkfd_pkg
R = rkfd_pkg.load()
from oracle.pyoracle import Oracle
plugin = sys.argv[1] if len(sys.argv) > 1 else "volume"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 140
resync = len(sys.argv) > 3 and sys.argv[3] == "resync"
M = R.scenarios.MODELS
w = R.World(solver=R.SOLVER_MLCP if plugin == "mlcp" else R.SOLVER_VOLUME); w.contact_info(os.path.join(M, "contactinfo.ztk"))
a = w.reg_file(os.path.join(M, "arm_revroot.ztk")); wl = w.reg_file(os.path.join(M, "wall.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
w.pair_chain_unreg(a)
m = w.model.contents
dis = np.zeros(m.ndof); dis[0] = 1.40; dis[1] = 0.12
b = R.Batch(w, 1, max_rigid=7 if plugin == "volume" else 16)
b.set_state(dis[None], np.zeros((1, m.ndof))); b.update_init()
o = Oracle(w.model); o.set_state(dis, np.zeros(m.ndof)); o.update_init()
inp = np.zeros(m.nlink); t = 0.0; tc = 0.0; target = (1.9, 0.12)
for k in range(nsteps):
    if tc <= t + 1e-9:
        d, v, _a = o.get_state()
        for i in range(2):
            inp[w.link_offset(a) + i] = -60.0 * (d[i] - target[i]) - 3.0 * v[i]
        o.set_motor_input(inp); b.set_motor_input(inp[None]); tc += 0.002
    if resync:
        d, v, _ = o.get_state(); b.set_state(d[None], v[None]); b.set_broken(o.get_broken()[None]); b.set_pivot(*[x[None] for x in o.get_pivot()])
        if plugin == "mlcp": b.set_contact(*[x[None] for x in o.get_contact()[:3]])
    o.update(); b.update(1); t += 0.001
    st = b.status()
    d, v, ac = b.get_state(); od, ov, oa = o.get_state()
    e = (np.abs(d[0] - od).max(), np.abs(v[0] - ov).max(), np.abs(ac[0] - oa).max() / max(1, np.abs(oa).max()))
    if e[2] > 1e-9 or st != 0 or (b.get_broken()[0] != o.get_broken()).any() or k % 20 == 0:
        print(k + 1, "status", st, "broken", b.get_broken()[0].tolist(), o.get_broken().tolist(), "pairs", len(o.volume_pairs()) if plugin == "volume" else int(o.get_contact()[0].sum()),
              "dis %.1e vel %.1e acc %.1e" % e, flush=True)
