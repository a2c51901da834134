"""debug aid: one step of the rolling-cylinder world from a fixed state on the GPU against the oracle"""
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
dis = np.array([0.004463592430925724, -1.0169736572512807e-20, 0.039858397096680924, 1.261781843613411e-18, 0.07819171288705688, 4.025209050850159e-19])
vel = np.array([0.2589552989568131, -1.3757220976972952e-18, -0.018274073699851533, -4.727212925616903e-16, 6.054735459501072, -1.0880093285970527e-16])
np.set_printoptions(precision=12, linewidth=220)
bt = R.Batch(w, 1, max_rigid=1)
bt.set_state(dis[None, :], vel[None, :]); bt.update_init()
o = Oracle(w.model); o.set_state(dis, vel); o.update_init()
print("eval gpu", bt.get_state()[2][0]); print("eval orc", o.get_state()[2])
bt.update(1); o.update()
print("step gpu", bt.get_state()[2][0], bt.status()); print("step orc", o.get_state()[2])
print("step dis diff", bt.get_state()[0][0] - o.get_state()[0]); print("step vel diff", bt.get_state()[1][0] - o.get_state()[1])
gd, gv = bt.get_state()[0][0].copy(), bt.get_state()[1][0].copy()
bt2 = R.Batch(w, 1, max_rigid=1); bt2.set_state(gd[None, :], gv[None, :]); bt2.update_init(); print("fresh eval at the GPU's own post-step state", bt2.get_state()[2][0])
o9 = Oracle(w.model); o9.set_state(gd, gv); o9.update_init(); print("oracle eval at the GPU's own post-step state", o9.get_state()[2])
# the stage states of the RKG step, evaluated one by one (mode 2: no commit)
d1 = np.array([0.004721609766877518, -1.0707984873895913e-20, 0.03983922258211339, 8.353164171936729e-19, 0.0843094440998165, 2.780025519651297e-19])
v1 = np.array([0.25689425839221347, -1.9892488426118368e-18, -0.02009386952139051, -2.6069697427611444e-16, 6.190261680127373, -6.890497652592642e-17])
bt.set_state(d1[None, :], v1[None, :]); bt.update_init()
o.set_state(d1, v1); o.update_init()
print("post-step state: eval gpu", bt.get_state()[2][0]); print("post-step state: eval orc", o.get_state()[2])
for s in (1e-13, 1e-10, 1e-7):
    rng = np.random.default_rng(1)
    out = []
    for t in range(12):
        d2 = d1 * (1 + rng.uniform(-1, 1, 6) * s); v2 = v1 * (1 + rng.uniform(-1, 1, 6) * s)
        bt.set_state(d2[None, :], v2[None, :]); bt.update_init(); out.append(round(float(bt.get_state()[2][0][0]), 3))
    print("gpu acc[0] under perturbations of", s, sorted(set(out)), [out.count(x) for x in sorted(set(out))])
print("---- one step from perturbed start states")
for s in (1e-14, 1e-12, 1e-9):
    rng = np.random.default_rng(2)
    og, oo = [], []
    for t in range(16):
        d2 = dis * (1 + rng.uniform(-1, 1, 6) * s); v2 = vel * (1 + rng.uniform(-1, 1, 6) * s)
        bt.set_state(d2[None, :], v2[None, :]); bt.update_init(); bt.update(1); og.append(round(float(bt.get_state()[2][0][0]), 3))
        o.set_state(d2, v2); o.update_init(); o.update(); oo.append(round(float(o.get_state()[2][0]), 3))
    print("perturbation", s, "gpu", sorted(set(og)), [og.count(x) for x in sorted(set(og))], "oracle", sorted(set(oo)), [oo.count(x) for x in sorted(set(oo))])
