"""timing sweep over batch size and steps-per-launch (diagnostic; not the bench)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import rkfd_pkg
R = rkfd_pkg.load()
import torch

name = sys.argv[1] if len(sys.argv) > 1 else "config4"
for B in (1024, 2048, 4096, 8192, 16384):
    sc = R.scenarios.CONFIGS[name](batch=B)
    b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    b.set_state(sc["dis"], sc["vel"]); b.update_init(); b.update(20); assert b.status() == 0
    for per in (1, 10, 100):
        n = 100 // per
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            b.update(per)
        b.status(); t1 = time.perf_counter()
        print(f"{name} B={B} steps/launch={per}: {(t1-t0)/100*1e3:.4f} ms/step  {B*100/(t1-t0)/1e6:.2f} M steps/s", flush=True)
