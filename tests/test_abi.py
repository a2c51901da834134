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


def test_node_level_has_no_cpu_fallback_either(R):
    """rkfdNodeCreate (all the GPUs of a node from one process, include/rkfd_hip.h) fails loudly without a GPU"""
    if R.lib().rkfdHipDeviceCount() > 0:
        return
    sc = R.scenarios.config2(batch=4)
    try:
        R.Node(sc["world"], 4, max_rigid=0)
    except R.RkfdError as e:
        assert "no HIP device" in str(e)
        return
    raise AssertionError("rkfdNodeCreate must fail without a GPU")


def test_lds_budget_and_limits_on_host(R):
    """rkfdLdsBytesFor works without a GPU: the humanoid workloads fit 8 workgroups per CU
    (<= 20 480 B of LDS per instance), oversize worlds are rejected with a message"""
    import os
    L = R.lib()
    for cfg, cap in (("config2", 0), ("config3", 0), ("config4", 8)):
        sc = R.scenarios.CONFIGS[cfg](batch=1)
        n = L.rkfdLdsBytesFor(sc["world"].model, cap)
        assert 0 < n <= 20480, (cfg, n)
    w = R.World()
    for _ in range(3):
        w.reg_file(os.path.join(R.scenarios.MODELS, "chain30.ztk"))
    assert L.rkfdLdsBytesFor(w.model, 0) < 0
    assert b"exceeds" in L.rkfdHipLastError()
