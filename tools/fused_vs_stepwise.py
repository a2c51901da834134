"""rkfdBatchUpdate(n) as ONE launch of n fused steps against n launches of one step (diagnostic).
A fused launch keeps the state in LDS across steps, but when the batch exceeds the resident slots its second round
only starts after n steps of the first; per-step launches on the internal streams keep the slots full."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rkfd_pkg
R = rkfd_pkg.load()
name = sys.argv[1] if len(sys.argv) > 1 else "config4"
for B in (2816, 4096, 16384):
    sc = R.scenarios.CONFIGS[name](batch=B)
    for split, fused in ((1, True), (3, True), (1, False), (3, False)):
        b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
        b.set_split(split)
        b.set_state(sc["dis"], sc["vel"]); b.update_init()
        b.update(100); b.status()                       # contacts built up
        t0 = time.time()
        if fused:
            b.update(200)
        else:
            for _ in range(200):
                b.update(1)
        st = b.status()
        dt = time.time() - t0
        print(f"{name} batch {B:5d} split {split} {'one call, 200 steps' if fused else '200 calls of one step'}: {B*200/dt/1e6:7.3f} M steps/s  status {st}", flush=True)
        del b
