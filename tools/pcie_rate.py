#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload: the bench's `value` is measured with the state resident in
HBM; here every rollout of H steps also uploads its start states (rkfdBatchSetState: host -> device) and downloads the final
ones (rkfdBatchGetState: device -> host, {dis, vel, acc}), as a caller without device-resident start states would.
usage: python tools/pcie_rate.py [workload] [horizon] [rollouts]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rkfd_pkg
R = rkfd_pkg.load()
name = sys.argv[1] if len(sys.argv) > 1 else "config4"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 25
N = int(sys.argv[3]) if len(sys.argv) > 3 else 400
B = 4096
sc = R.scenarios.CONFIGS[name](batch=B)
b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
b.specialize()
b.set_state(sc["dis"], sc["vel"]); b.set_split(3); b.update_init()
b.tune_instances_per_wave(H)
b.snapshot()
dis = np.ascontiguousarray(sc["dis"]); vel = np.ascontiguousarray(sc["vel"])
for mode in ("resident", "pcie"):
    for rep in range(2):                      # first pass warms up
        t0 = time.time()
        for r in range(N):
            b.restore()                       # contact / pivot state of the start (device side, as in bench.py)
            if mode == "pcie":
                b.set_state(dis, vel)         # host -> device: 2 x B x ndof doubles
            b.update(H)
            if mode == "pcie":
                b.get_state()                 # device -> host: 3 x B x ndof doubles (synchronises)
        assert b.status() == 0
        dt = time.time() - t0
    print("%-9s %d rollouts of %d steps x %d instances: %.3f ms per rollout, %.3f M steps/s" % (mode, N, H, B, 1e3 * dt / N, B * H * N / dt / 1e6), flush=True)
print("bytes per rollout over PCIe: %.2f MB up, %.2f MB down" % (2 * dis.nbytes / 1e6, 3 * dis.nbytes / 1e6))
