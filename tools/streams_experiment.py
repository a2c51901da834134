"""does splitting the batch over K streams hide the tail of each step? (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import rkfd_pkg
R = rkfd_pkg.load()
name = sys.argv[1] if len(sys.argv) > 1 else "config4"
B = 4096
sc = R.scenarios.CONFIGS[name](batch=B)
for K in (1, 2, 4, 8):
    n = B // K
    streams = [torch.cuda.Stream() for _ in range(K)]
    bs = []
    for k in range(K):
        b = R.Batch(sc["world"], n, max_rigid=sc["max_rigid"]); b.set_state(sc["dis"][k*n:(k+1)*n], sc["vel"][k*n:(k+1)*n]); b.update_init(streams[k].cuda_stream); bs.append(b)
    for _ in range(200):
        for k in range(K): bs[k].update(1, streams[k].cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        for k in range(K): bs[k].update(1, streams[k].cuda_stream)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = max(b.status(s.cuda_stream) for b, s in zip(bs, streams))
    print(f"{name}: {K} stream(s) x {n} instances: {dt/200*1e3:.4f} ms/step  {B*200/dt/1e6:.2f} M steps/s  status {st}", flush=True)
