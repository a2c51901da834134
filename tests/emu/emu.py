"""ctypes wrapper of the lane emulator (tests/emu/rkfd_emu.cpp): runs the DEVICE code of
roki-fd_amd/csrc/rkfd_device.h on host threads.  Development / test harness only."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB_PATH = os.environ.get("RKFD_EMU_LIB", os.path.join(HERE, "librkfd_emu.so"))      # (RKFD_EMU_LIB: a sanitizer build, tools/emu_asan.sh)
_lib = None
_lib_w2 = None
LIB_PATH_W2 = os.environ.get("RKFD_EMU_LIB_W2", os.path.join(HERE, "librkfd_emu_w2.so"))      # (two instances per wavefront; RKFD_EMU_LIB_W2: its sanitizer build)


class DevState(C.Structure):
    _fields_ = [("dis", C.c_void_p), ("vel", C.c_void_p), ("acc", C.c_void_p), ("motor_in", C.c_void_p),
                ("piv_type", C.c_void_p), ("piv_prev", C.c_void_p), ("cv_active", C.c_void_p), ("cv_type", C.c_void_p),
                ("cv_ref", C.c_void_p), ("cv_f", C.c_void_p), ("brk", C.c_void_p), ("stat", C.c_void_p), ("prof", C.c_void_p), ("dbg", C.c_void_p), ("dbg_stride", C.c_int), ("batch", C.c_int)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.run(["make", "-C", ROOT, "emu"], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(LIB_PATH)
        L.rkfd_emu_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(DevState), C.c_int, C.c_int]
        _lib = L
    return _lib


def lib_w2():
    """the harness built with RKFD_W = 2: two instances per emulated wavefront, 32 lanes each"""
    global _lib_w2
    if _lib_w2 is None:
        if not os.path.exists(LIB_PATH_W2):
            subprocess.run(["make", "-C", ROOT, "emu"], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(LIB_PATH_W2)
        L.rkfd_emu_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(DevState), C.c_int, C.c_int]
        _lib_w2 = L
    return _lib_w2


class EmuBatch:
    """Same surface as roki_fd_amd.Batch, backed by the emulator."""

    def __init__(self, world, batch, max_rigid=8, ipw=1):
        self.world = world
        self.ipw = ipw                      # instances per emulated wavefront (1 or 2)
        m = world.model.contents
        self.B, self.ndof, self.nlink, self.ncand = batch, m.ndof, m.nlink, m.ncand
        self.max_rigid = max_rigid
        B = batch
        self.dis = np.zeros((B, self.ndof)); self.vel = np.zeros((B, self.ndof)); self.acc = np.zeros((B, self.ndof))
        self.motor_in = np.zeros((B, self.nlink))
        self.piv_type = np.zeros((B, self.nlink), dtype=np.int32); self.piv_prev = np.zeros((B, self.nlink))
        self.cv_active = np.zeros((B, max(self.ncand, 1)), dtype=np.int32); self.cv_type = np.zeros_like(self.cv_active)
        self.cv_ref = np.zeros((B, max(self.ncand, 1), 3)); self.cv_f = np.zeros_like(self.cv_ref)
        self.brk = np.zeros((B, self.nlink), dtype=np.int32)
        self.dbg = np.zeros((B, 18 * self.nlink))
        self.err = 0

    def _run(self, mode, nsteps):
        st = DevState()
        for k in ("dis", "vel", "acc", "motor_in", "piv_type", "piv_prev", "cv_active", "cv_type", "cv_ref", "cv_f", "brk", "dbg"):
            setattr(st, k, getattr(self, k).ctypes.data)
        st.dbg_stride = 18 * self.nlink
        st.batch = self.B
        self.err = (lib_w2() if self.ipw == 2 else lib()).rkfd_emu_run(C.cast(self.world.model, C.c_void_p), self.max_rigid, C.byref(st), mode, nsteps)
        if self.err < 0:
            raise RuntimeError("emulator: device model build failed")

    def set_state(self, dis, vel):
        self.dis[...] = np.asarray(dis).reshape(self.B, self.ndof); self.vel[...] = np.asarray(vel).reshape(self.B, self.ndof)

    def get_state(self):
        return self.dis.copy(), self.vel.copy(), self.acc.copy()

    def set_motor_input(self, inp):
        self.motor_in[...] = np.asarray(inp).reshape(self.B, self.nlink)

    def get_contact(self):
        n = self.ncand
        # like rkfdBatchGetContact: a candidate out of contact has no state, report zeros
        act = self.cv_active[:, :n].copy(); on = act != 0
        return act, self.cv_type[:, :n] * on, self.cv_ref[:, :n] * on[:, :, None], self.cv_f[:, :n] * on[:, :, None]

    def set_contact(self, act, typ, ref):
        n = self.ncand
        self.cv_active[:, :n] = np.asarray(act).reshape(self.B, n); self.cv_type[:, :n] = np.asarray(typ).reshape(self.B, n)
        self.cv_ref[:, :n] = np.asarray(ref).reshape(self.B, n, 3)

    def set_pivot(self, typ, prev):
        self.piv_type[...] = np.asarray(typ).reshape(self.B, self.nlink); self.piv_prev[...] = np.asarray(prev).reshape(self.B, self.nlink)

    def get_pivot(self):
        return self.piv_type.copy(), self.piv_prev.copy()

    def get_broken(self):
        return self.brk.copy()

    def set_broken(self, broken):
        self.brk[...] = np.asarray(broken).reshape(self.B, self.nlink)

    def update_init(self):
        self._run(1, 0)

    def update(self, nsteps=1):
        self._run(0, nsteps)

    def eval(self, do_up_ref=False):
        self._run(1 if do_up_ref else 2, 0)

    def status(self):
        return self.err
