"""histogram of active contacts per instance over a run (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rkfd_pkg, numpy as np
R = rkfd_pkg.load()
name = sys.argv[1] if len(sys.argv) > 1 else "config4"
sc = R.scenarios.CONFIGS[name](batch=4096)
b = R.Batch(sc["world"], 4096, max_rigid=sc["max_rigid"])
b.set_state(sc["dis"], sc["vel"]); b.update_init()
for n in (0, 20, 50, 100, 200):
    if n: b.update(n)
    act = b.get_contact()[0].sum(1)
    print(name, "after +%d steps:" % n, np.bincount(act, minlength=9).tolist(), "status", b.status(), flush=True)
