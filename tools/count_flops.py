"""Algorithmic flops per instance-step ("an instrumented count from the CPU restatement"):
runs the flop-counting build of the oracle (oracle/flopcount.cpp) on a few instances of every config
and writes profiles/<tag>_flops.json.  The count covers the window bench.py times: one rollout of `horizon` steps from
the scenario's start states (the standing states of configs 3 / 4 / 5).  CPU only.
usage: python3 tools/count_flops.py [tag] [horizon]"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import rkfd_pkg

R = rkfd_pkg.load()
import oracle.pyoracle as po          # noqa: E402
from oracle.pyoracle import Oracle     # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "count"], check=True, stdout=subprocess.DEVNULL)
po._lib = None
po.LIB_PATH = os.path.join(ROOT, "oracle", "_build", "librkfd_oracle_count.so")
L = po.lib()
L.rkfdOracleFlops.restype = C.c_ulonglong
out = {}
for name in ("config1", "config2", "config3", "config4", "config4v", "config5"):
    sc = R.scenarios.CONFIGS[name](batch=4)
    tot = 0
    for i in range(sc["dis"].shape[0]):
        o = Oracle(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init()
        L.rkfdOracleFlopsReset()
        o.update_n(nsteps)
        tot += L.rkfdOracleFlops()
        o.close()
    out[name] = {"flops_per_instance_step": tot / (nsteps * sc["dis"].shape[0]), "instances": int(sc["dis"].shape[0]), "steps": nsteps, "horizon": nsteps,
                 "note": "+ - * / sqrt sin cos exp atan2 = 1 flop each, counted in oracle/rkfd_oracle.c (link-local ABA, column-probed MLCP)"}
    print(name, "%.0f flops per instance-step" % out[name]["flops_per_instance_step"], flush=True)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_flops.json"), "w"), indent=1)
