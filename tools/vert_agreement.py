"""How often does the device's Vert QP path leave the oracle's?  N random box drops (tilted, sliding, spinning)
onto the rigid floor, S steps; an instance 'agrees' while contact sets, stick/slip types and velocities (1e-6)
match.  Diagnostic for the knife-edge decisions of the active-set method (1e-12 absolute tests).
usage: python3 tools/vert_agreement.py [N] [S] [pyramid] [vert|mlcp]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rkfd_pkg
R = rkfd_pkg.load()
from oracle.pyoracle import Oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = int(sys.argv[2]) if len(sys.argv) > 2 else 60
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
PLUG = sys.argv[4] if len(sys.argv) > 4 else "vert"
M = R.scenarios.MODELS
w = R.World(solver=R.SOLVER_VERT if PLUG == "vert" else R.SOLVER_MLCP); w.contact_info(os.path.join(M, "contactinfo.ztk")); w.set_pyramid(P)
w.reg_file(os.path.join(M, "box.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
m = w.model.contents
rng = np.random.default_rng(11)
dis = np.zeros((N, 6)); vel = np.zeros((N, 6))
dis[:, 3:6] = rng.uniform(-0.4, 0.4, (N, 3)) * (rng.random((N, 1)) < 0.8)
vel[:, 0:2] = rng.uniform(-0.5, 0.5, (N, 2)); vel[:, 3:6] = rng.uniform(-1.5, 1.5, (N, 3)) * (rng.random((N, 1)) < 0.5)
for i in range(N):
    dis[i, 2] = 0.2
    dis[i, 2] = 0.2 - R.scenarios.lowest_vertex_z(m, dis[i], 0) - 0.0002      # lowest vertex 0.2 mm inside the floor
cap = min(8, 64 // P)
b = R.Batch(w, N, max_rigid=cap); b.set_state(dis, vel); b.update_init()
orc = []
for i in range(N):
    o = Oracle(w.model); o.set_state(dis[i], vel[i]); o.update_init(); orc.append(o)
alive = np.ones(N, dtype=bool); first_bad = np.full(N, -1); contact_steps = 0
for s in range(1, S + 1):
    b.update(1); d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
    for i, o in enumerate(orc):
        o.update()
        if not alive[i]:
            continue
        od, ov, oa = o.get_state(); oact, otyp, _, of = o.get_contact(); contact_steps += int(oact.sum() > 0)
        ok = (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all() and np.abs(v[i] - ov).max() < 1e-6 * max(1, np.abs(ov).max())
        if not ok:
            alive[i] = False; first_bad[i] = s
print(f"{PLUG} pyramid {P}: {alive.sum()}/{N} instances agree through {S} steps; status {b.status()}; steps with contact (while agreeing): {contact_steps}; "
      f"first disagreements at steps {sorted(first_bad[first_bad > 0].tolist())[:12]} (instances {np.argsort(np.where(first_bad > 0, first_bad, 10**9))[:int((first_bad > 0).sum())][:12].tolist()}); oracle cycle stops total {sum(o.qp_cycle_stops() for o in orc)}")
