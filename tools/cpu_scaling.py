#!/usr/bin/env python3
"""How the CPU oracle scales over OS threads on this host (the all-cores leg of bench.py's cpu_baseline): steps/s of config 4's
rollouts at 1, 2, 4 ... threads (pthreads inside the oracle library), next to what the job may use (affinity mask, cgroup quota).
usage: python tools/cpu_scaling.py [workload] [seconds per point]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rkfd_pkg
R = rkfd_pkg.load()
from oracle import pyoracle
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "config4"
sec = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
print("cpu share:", bench.cpu_share())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/self/cgroup"):
    try:
        print(f, "=", open(f).read().strip().replace("\n", " | "))
    except OSError as e:
        print(f, ":", e.strerror)
aff = len(os.sched_getaffinity(0))
sc = R.scenarios.CONFIGS[wl](batch=64)
one = None
n = 1
pts = []
while n <= aff:
    pts.append(n); n *= 2
if pts[-1] != aff:
    pts.append(aff)
for n in pts:
    steps, dt = pyoracle.rollouts_mt(sc["world"].model, sc["dis"], sc["vel"], 25, sec, nthreads=n)
    v = steps / dt
    one = one or v
    print("%4d threads: %9.0f steps/s = %5.1f x one thread" % (n, v, v / one), flush=True)
