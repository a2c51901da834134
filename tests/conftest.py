import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def R():
    """the product package (roki-fd_amd) with its library built"""
    import subprocess
    import rkfd_pkg
    mod = rkfd_pkg.load()
    if not os.path.exists(mod.LIB_PATH):
        subprocess.run(["make", "-C", ROOT, "all"], check=True, stdout=subprocess.DEVNULL)
    mod.lib()
    return mod


@pytest.fixture(scope="session")
def oracle_cls():
    from oracle.pyoracle import Oracle, lib
    lib()
    return Oracle
