"""host time to enqueue a step (three launches + events through ctypes) against the GPU time of the step (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rkfd_pkg
R = rkfd_pkg.load()
sc = R.scenarios.CONFIGS["config4"](batch=4096)
b = R.Batch(sc["world"], 4096, max_rigid=sc["max_rigid"]); b.set_split(3)
b.set_state(sc["dis"], sc["vel"]); b.update_init(); b.update(20); b.status()
t0 = time.perf_counter()
for _ in range(200):
    b.update(1)
t1 = time.perf_counter()
b.status()
t2 = time.perf_counter()
print("enqueue of 200 steps: %.1f us per step on the host; until the GPU is done: %.1f us per step" % ((t1-t0)/200*1e6, (t2-t0)/200*1e6))
