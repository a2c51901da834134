"""long-run sanity (diagnostic): many steps at the full batch, status and finiteness checks"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rkfd_pkg
R = rkfd_pkg.load()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for name in sys.argv[2:] or ["config4", "config4v", "config3", "config5"]:
    sc = R.scenarios.CONFIGS[name](batch=4096)
    b = R.Batch(sc["world"], 4096, max_rigid=sc["max_rigid"])
    kern = "generic kernel"
    if b.lds_bytes <= 64 * 1024:          # the path the bench times: world-specific kernel, lane mapping by measurement
        b.specialize(); kern = "world-specific kernel"
    b.set_state(sc["dis"], sc["vel"]); b.update_init()
    if b.lds_bytes <= 64 * 1024:
        kern += ", %d instance(s) per wavefront" % b.tune_instances_per_wave(10)[0]
    t0 = time.time(); worst = 0
    for k in range(steps // 100):
        b.update(100); st = b.status(); worst = max(worst, st)
    d, v, a = b.get_state(); act = b.get_contact()[0]
    print(f"{name} ({kern}): {steps} steps x 4096 in {time.time()-t0:.1f} s, worst status {worst}, finite {bool(np.isfinite(d).all() and np.isfinite(v).all() and np.isfinite(a).all())}, "
          f"max |vel| {np.abs(v).max():.2f}, base height min/max {d[:, sc['dis'].shape[1]-sc['dis'].shape[1]+2].min():.3f}/{d[:, 2].max():.3f}, mean contacts {act.sum(1).mean():.2f}", flush=True)
