"""more of tests/test_random_trees.py (diagnostic): random tree topologies, GPU path against the oracle, with other
seeds and more trees than the test tier affords.  usage: python tools/stress_random.py [seed_offset] [scale]"""
import os, sys, tempfile, pathlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rkfd_pkg
from oracle.pyoracle import Oracle, lib as _olib
import test_random_trees as T
R = rkfd_pkg.load()
off = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tmp = pathlib.Path(tempfile.mkdtemp())
_olib(); O = Oracle
worst = {"free": 0.0, "motors": 0.0, "contacts": 0.0}
rng = np.random.default_rng(off)
n = 0
for k in range(60 * scale):                      # free motion
    seed = off + k
    nlink = int(rng.integers(3, 49)); root = ["float", "fixed", "revolute"][k % 3]
    w, _ = T._world(R, tmp, seed, nlink, root)
    m = w.model.contents
    if m.ndof > 64 or m.ndof == 0:      # (a tree of fixed joints only has no state to compare)
        continue
    dis, vel = T._state(w, seed, 4)
    b = R.Batch(w, 4, max_rigid=0); b.set_state(dis, vel); b.update_init(); b.update(3)
    assert b.status() == 0
    d, v, a = b.get_state()
    for i in range(4):
        o = O(w.model); o.set_state(dis[i], vel[i]); o.update_init(); o.update_n(3)
        for x, y in zip((d[i], v[i], a[i]), o.get_state()):
            e = np.abs(x - y).max() / max(1.0, np.abs(y).max()); worst["free"] = max(worst["free"], e)
            assert e < 1e-8, ("free", seed, nlink, root, e)
    n += 1
print("free motion: %d trees, worst rel. error %.2e" % (n, worst["free"]), flush=True)
n = 0
for k in range(30 * scale):                      # motors and joint friction
    seed = off + 5000 + k
    nlink = int(rng.integers(3, 30)); root = ["float", "fixed", "revolute"][k % 3]
    w, _ = T._world(R, tmp, seed, nlink, root, motors=True)
    m = w.model.contents
    dis, vel = T._state(w, seed, 4); vel *= 0.3
    inp = np.random.default_rng(seed).uniform(-30, 30, (4, m.nlink))
    b = R.Batch(w, 4, max_rigid=0); b.set_state(dis, vel); b.set_motor_input(inp); b.update_init(); b.update(20)
    assert b.status() == 0
    d, v, a = b.get_state(); pt, pp = b.get_pivot(); mt = m.arr("mtype", m.nlink)
    for i in range(4):
        o = O(w.model); o.set_state(dis[i], vel[i]); o.set_motor_input(inp[i]); o.update_init(); o.update_n(20)
        for x, y in zip((d[i], v[i], a[i]), o.get_state()):
            e = np.abs(x - y).max() / max(1.0, np.abs(y).max()); worst["motors"] = max(worst["motors"], e)
            assert e < 1e-7, ("motors", seed, nlink, root, e)
        assert (pt[i][mt == 2] == o.get_pivot()[0][mt == 2]).all(), ("pivot types", seed)
    n += 1
print("motors: %d trees, worst rel. error %.2e" % (n, worst["motors"]), flush=True)
n = 0; ncontact = 0; nsplit = 0
for k in range(20 * scale):                      # falling onto the rigid floor, MLCP
    seed = off + 9000 + k
    nlink = int(rng.integers(4, 20))
    w, h = T._world(R, tmp, seed, nlink, "float", shapes=min(4, nlink), floor=True)
    m = w.model.contents
    B = 4
    dis = np.zeros((B, m.ndof)); vel = np.zeros((B, m.ndof))
    r2 = np.random.default_rng(seed)
    dis[:, 6:] = r2.uniform(-0.5, 0.5, (B, m.ndof - 6)); dis[:, 3:6] = r2.uniform(-0.3, 0.3, (B, 3))
    vel[:, 2] = -0.3; vel[:, 3:6] = r2.uniform(-1.0, 1.0, (B, 3))
    for i in range(B):
        dis[i, 2] -= R.scenarios.lowest_vertex_z(m, dis[i], h) - 0.002
    b = R.Batch(w, B, max_rigid=16); b.set_state(dis, vel); b.update_init()
    orc = []
    for i in range(B):
        o = O(w.model); o.set_state(dis[i], vel[i]); o.update_init(); orc.append(o)
    alive = [True]*B
    for chunk in range(6):
        b.update(10); assert b.status() == 0, seed
        d, v, a = b.get_state(); act = b.get_contact()[0]
        for i, o in enumerate(orc):
            o.update_n(10)
            if not alive[i]:
                continue
            od, ov, oa = o.get_state(); oact = o.get_contact()[0]
            if not (act[i] == oact).all():       # a make/break decision fell the other way: count, stop comparing this one
                alive[i] = False; nsplit += 1; continue
            tol = 1e-8 if chunk == 0 else 1e-6
            for x, y in ((d[i], od), (v[i], ov)):
                e = np.abs(x - y).max() / max(1.0, np.abs(y).max()); worst["contacts"] = max(worst["contacts"], e)
                assert e < tol, ("contacts", seed, nlink, chunk, e)
            ncontact += int(oact.sum())
    n += 1
print("contacts: %d trees x 4, %d contact-steps compared, %d instances left the oracle's contact sets, worst rel. error %.2e" % (n, ncontact, nsplit, worst["contacts"]), flush=True)
print("stress ok")
