"""Imports the package directory ``roki-fd_amd`` (not a valid Python identifier) as
module ``roki_fd_amd``."""
import importlib.util
import os
import sys

_NAME = "roki_fd_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "roki-fd_amd")
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(root, "__init__.py"), submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


def device_source_hash():
    """sha256 over the sources the step kernels are built from (device headers, the C-ABI / launch file, the host-built tables):
    the stamp that ties a rocprofv3 summary under profiles/ to the code it was taken on.  bench.py reports counter-derived
    figures (HBM traffic, VALU instructions per instance-step) only from a profile whose stamp equals the current tree's."""
    import glob
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    pk = os.path.join(root, "roki-fd_amd", "csrc")
    files = [os.path.join(root, "include", "rkfd_model.h")] + sorted(glob.glob(os.path.join(pk, "*.h"))) + sorted(glob.glob(os.path.join(pk, "device", "*.h"))) \
        + [os.path.join(pk, "rkfd_capi.hip"), os.path.join(pk, "rkfd_devmodel.cpp")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode()); h.update(b"\0")
        with open(f, "rb") as fp:
            h.update(fp.read())
    return h.hexdigest()[:16]


def git_head():
    """short commit id of the tree, None where there is no repository (the GPU box gets a snapshot without .git)"""
    import subprocess
    try:
        r = subprocess.run(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10)
        return r.stdout.strip() or None if r.returncode == 0 else None
    except (OSError, subprocess.SubprocessError):
        return None
