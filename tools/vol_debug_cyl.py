"""debug aid: finds the steps where the GPU and the oracle disagree on the rolling-cylinder scenario and prints the start states"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rkfd_pkg
R = rkfd_pkg.load()
from oracle.pyoracle import Oracle
M = R.scenarios.MODELS
w = R.World(solver=R.SOLVER_VOLUME)
w.contact_info(os.path.join(M, "contactinfo.ztk"))
w.reg_file(os.path.join(M, "cylinder.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
B = 4
dis = np.zeros((B, 6)); vel = np.zeros((B, 6))
for b in range(B):
    dis[b, 2] = 0.04 - 1e-5; dis[b, 5] = 0.3 * b; vel[b, 0] = 0.3 * np.cos(0.3 * b); vel[b, 1] = 0.3 * np.sin(0.3 * b)
    vel[b, 3] = -0.5 * 7.5 * np.sin(0.3 * b); vel[b, 4] = 0.5 * 7.5 * np.cos(0.3 * b)
bt = R.Batch(w, B, max_rigid=1)
os_ = []
for b in range(B):
    o = Oracle(w.model); o.set_state(dis[b], vel[b]); o.update_init(); os_.append(o)
np.set_printoptions(precision=17, linewidth=250)
for k in range(300):
    sd = np.array([o.get_state()[0] for o in os_]); sv = np.array([o.get_state()[1] for o in os_])
    bt.set_state(sd, sv); bt.update_init(); bt.update(1)
    d, v, a = bt.get_state()
    for b, o in enumerate(os_):
        o.update()
        od, ov, oa = o.get_state()
        e = max(np.abs(d[b] - od).max(), np.abs(v[b] - ov).max(), np.abs(a[b] - oa).max() / max(1.0, np.abs(oa).max()))
        if e > 1e-7:
            print("step", k, "inst", b, "err", e, "status", bt.status())
            print(" dis", repr(sd[b].tolist())); print(" vel", repr(sv[b].tolist()))
            print(" pairs", [(len(p["planes"]), p["type"], p["wrench"].tolist()) for p in o.volume_pairs()])
