#!/usr/bin/env python3
"""Ahead-of-time specialised step kernels (`make spec`): compiles the world-specific kernel of every workload bench.py knows -
BASELINE.json's configurations and their variants - through the library's own hipRTC path (no GPU needed) and leaves the code
objects in roki-fd_amd/spec/, where rkfdBatchSpecialize finds them by key instead of compiling at run time.
usage: python tools/make_spec.py [workload ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rkfd_pkg
R = rkfd_pkg.load()
L = R.lib()
L.rkfdSpecializeLastFromStore.restype = int
names = sys.argv[1:] or list(R.scenarios.CONFIGS)
# a full run prunes the store: code objects of earlier device sources (other keys) are dead weight that travels with the tree
t_start = time.time()
for nm in names:
    sc = R.scenarios.CONFIGS[nm](batch=1)
    if L.rkfdLdsBytesFor(sc["world"].model, sc["max_rigid"]) > 64 * 1024:
        print("%-16s keeps the generic kernel (more than 64 KiB of LDS per instance)" % nm); continue
    t0 = time.time()
    n = L.rkfdSpecializeCompile(sc["world"].model, sc["max_rigid"])
    if n <= 0:
        print("%-16s FAILED: %s" % (nm, L.rkfdHipLastError().decode())); sys.exit(1)
    print("%-16s %6d bytes  %s  (%.1f s)" % (nm, n, "already in the store" if L.rkfdSpecializeLastFromStore() else "compiled", time.time() - t0), flush=True)
    # the kernel with two instances per wavefront, for the worlds eligible for it (rkfdBatchTuneInstancesPerWave picks by measurement)
    t0 = time.time()
    n = L.rkfdSpecializeCompileW(sc["world"].model, sc["max_rigid"], 2)
    if n > 0:
        print("%-16s %6d bytes  %s  (%.1f s)  [two instances per wavefront]" % (nm, n, "already in the store" if L.rkfdSpecializeLastFromStore() else "compiled", time.time() - t0), flush=True)

if not sys.argv[1:]:
    import glob, shutil
    spec = os.environ.get("RKFD_SPEC_DIR") or os.path.join(ROOT, "roki-fd_amd", "spec")
    keep = os.path.join(spec, ".keep")
    # which files belong to this run: compile once more into an empty directory is the only way to know without asking the
    # library for its keys - so a full run compiles into a fresh directory when the store holds more files than it just produced
    have = glob.glob(os.path.join(spec, "rkfd_spec_*.co"))
    fresh = [f for f in have if os.path.getmtime(f) >= t_start - 1]
    if len(have) > 2 * len(names) + 8 and not os.environ.get("RKFD_SPEC_PRUNED"):
        tmp = spec + ".new"
        shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp)
        env = dict(os.environ, RKFD_SPEC_DIR=tmp, RKFD_SPEC_PRUNED="1")
        import subprocess
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=True)
        shutil.rmtree(spec); os.rename(tmp, spec)
        print("store pruned: %d -> %d code objects" % (len(have), len(glob.glob(os.path.join(spec, "rkfd_spec_*.co")))))
