#!/usr/bin/env python3
"""Register / scratch budget of the world-specific step kernels (rkfdBatchSpecialize), WITHOUT a GPU: compiles the kernel of each
world through hipRTC (rkfdSpecializeCompile), dumps the code object and reads the kernel's metadata note.
usage: python tools/spec_resources.py [world ...]        worlds: the names of scenarios.CONFIGS, arm_press, arm_fold, ball_roll, arm_spher"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rkfd_pkg
R = rkfd_pkg.load()
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def world(name):
    S = R.scenarios
    if name in S.CONFIGS:
        return S.CONFIGS[name](batch=1)
    if name.startswith("arm_press"):
        return S.arm_press(batch=1, root="revolute" if "rev" in name else "fixed", with_box="nobox" not in name)
    return getattr(S, name)(batch=1)


def resources(sc, ipw=1):
    """dict(vgpr, sgpr, scratch, vgpr_spill, sgpr_spill, lds) of the specialised kernel of scenario sc (ipw instances per wavefront)"""
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "k.co")
        os.environ["RKFD_SPEC_DUMP_CODE"] = co
        try:
            n = R.lib().rkfdSpecializeCompileW(sc["world"].model, sc["max_rigid"], ipw)
        finally:
            del os.environ["RKFD_SPEC_DUMP_CODE"]
        if n <= 0:
            raise RuntimeError(R.lib().rkfdHipLastError().decode())
        txt = subprocess.run([READELF, "--notes", co], check=True, capture_output=True, text=True).stdout
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, txt).group(1))
    return dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"), vgpr_spill=g("vgpr_spill_count"),
                sgpr_spill=g("sgpr_spill_count"), code_bytes=n)


if __name__ == "__main__":
    for nm in sys.argv[1:] or ["config2", "config3", "config4", "config4v", "config5", "arm_press", "arm_press_rev", "arm_fold", "config2:2", "config3:2", "config4:2"]:
        w_, _, ipw = nm.partition(":")
        print("%-14s %s" % (nm, resources(world(w_), int(ipw or 1))), flush=True)
