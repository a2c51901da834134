"""roki-fd_amd: MI355X-native batched rkFDUpdate path (host-side Python binding).

The product is the C-ABI shared library ``librkfd_amd.so`` (include/rkfd_hip.h,
include/roki_fd_amd.h).  This package is a thin ctypes binding over it, used by the
tests and by bench.py; it contains no numerics of its own and never falls back to a
CPU implementation: without the built library or without a GPU the device calls raise.
"""
from .binding import (  # noqa: F401
    LIB_PATH, lib, RkfdModel, World, Batch, Node, RkfdError,
    JOINT_FIXED, JOINT_REVOL, JOINT_PRISM, JOINT_FLOAT,
    SOLVER_VERT, SOLVER_MLCP, SOLVER_VOLUME, CONTACT_RIGID, CONTACT_ELASTIC, SF, KF,
)
from . import scenarios  # noqa: F401
from . import sharding  # noqa: F401
