"""The C-ABI library loads and exports every function include/*.h declares; without a GPU the
device entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = set()
    for h in ("rkfd_hip.h", "roki_fd_amd.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        # drop preprocessor lines together with their backslash continuations
        out, cont = [], False
        for line in txt.splitlines():
            if cont or line.lstrip().startswith("#"):
                cont = line.rstrip().endswith("\\")
                continue
            out.append(line)
        txt = "\n".join(out)
        for mm in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", txt):
            n = mm.group(1)
            if n.startswith(("rkfd", "rkFD", "rkChain", "rkJoint", "rkCD", "zVec")):
                names.add(n)
    return names


def test_exports(R):
    L = ctypes.CDLL(R.LIB_PATH)
    names = _declared_functions()
    assert len(names) > 40
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_no_cpu_fallback(R):
    L = R.lib()
    if L.rkfdHipDeviceCount() > 0:
        return      # on a GPU box the path is live; covered by the gpu tests
    sc = R.scenarios.config2(batch=2)
    try:
        R.Batch(sc["world"], 2)
    except R.RkfdError as e:
        assert "no HIP device" in str(e) or "HIP" in str(e)
        return
    raise AssertionError("rkfdBatchCreate must fail without a GPU")
