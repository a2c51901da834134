"""the world-specialised step kernel (rkfdBatchSpecialize) against the generic one on random trees (diagnostic):
free motion with motors, and falling onto the rigid floor under both plugins - states, contacts and pivots bit for bit"""
import os, sys, tempfile, pathlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rkfd_pkg
import test_random_trees as T
R = rkfd_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tmp = pathlib.Path(tempfile.mkdtemp())
rng = np.random.default_rng(4242)
done = 0
for k in range(n):
    seed = 70000 + k
    kind = k % 3
    if kind == 0:
        nlink = int(rng.integers(3, 40)); root = ["float", "fixed", "revolute"][(k // 3) % 3]
        w, h = T._world(R, tmp, seed, nlink, root, motors=True); mr = 0
    else:
        nlink = int(rng.integers(4, 18))
        w, h = T._world(R, tmp, seed, nlink, "float", shapes=min(4, nlink), floor=True,
                        solver=R.SOLVER_MLCP if kind == 1 else R.SOLVER_VERT); mr = 8
    m = w.model.contents
    if m.ndof > 64:
        continue
    B = 8
    dis, vel = T._state(w, seed, B); vel *= 0.3
    if kind:
        dis[:, :6] = 0; dis[:, 3:6] = np.random.default_rng(seed).uniform(-0.3, 0.3, (B, 3)); vel[:, :3] = 0; vel[:, 2] = -0.3
        for i in range(B):
            dis[i, 2] -= R.scenarios.lowest_vertex_z(m, dis[i], h) - 0.002
    inp = np.random.default_rng(seed).uniform(-10, 10, (B, m.nlink))
    out = []
    for spec in (False, True):
        b = R.Batch(w, B, max_rigid=mr)
        if spec:
            b.specialize()
        b.set_state(dis, vel); b.set_motor_input(inp); b.update_init(); b.update(30)
        st = b.status()
        out.append(b.get_state() + b.get_contact() + b.get_pivot() + (np.array([st]),))
    for x, y in zip(out[0], out[1]):
        assert np.array_equal(x, y), ("differs", seed, kind, nlink)
    done += 1
print("specialised = generic, bit for bit, on %d random worlds (motors / MLCP contacts / Vert contacts)" % done)
