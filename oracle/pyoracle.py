"""ctypes wrapper of the CPU oracle (oracle/rkfd_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "librkfd_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)


def _bind(path):
    L = C.CDLL(path)
    vp = C.c_void_p
    L.rkfdOracleCreate.argtypes = [vp]; L.rkfdOracleCreate.restype = vp
    L.rkfdOracleDestroy.argtypes = [vp]
    L.rkfdOracleSetState.argtypes = [vp, vp, vp]
    L.rkfdOracleGetState.argtypes = [vp, vp, vp, vp]
    L.rkfdOracleSetMotorInput.argtypes = [vp, vp]
    L.rkfdOracleTime.argtypes = [vp]; L.rkfdOracleTime.restype = C.c_double
    L.rkfdOracleGetContact.argtypes = [vp, vp, vp, vp, vp]
    L.rkfdOracleSetContact.argtypes = [vp, vp, vp, vp]
    L.rkfdOracleGetPivot.argtypes = [vp, vp, vp]
    L.rkfdOracleSetPivot.argtypes = [vp, vp, vp]
    L.rkfdOracleGetBroken.argtypes = [vp, vp]; L.rkfdOracleSetBroken.argtypes = [vp, vp]
    L.rkfdOracleUpdateInit.argtypes = [vp]
    L.rkfdOracleUpdate.argtypes = [vp]
    L.rkfdOracleUpdateN.argtypes = [vp, C.c_int]
    L.rkfdOracleLastQPIter.argtypes = [vp]
    L.rkfdOracleQPCycleStops.argtypes = [vp]
    L.rkfdOracleVolumeGuardHits.argtypes = [vp]
    L.rkfdOracleEval.argtypes = [vp, C.c_int]
    L.rkfdOracleGetLinkFrames.argtypes = [vp, vp, vp]
    L.rkfdOracleGetLinkVelAcc.argtypes = [vp, vp, vp]
    L.rkfdOracleGetMLCP.argtypes = [vp, vp, vp, vp, C.c_int]
    L.rkfdOracleVolumePairs.argtypes = [vp]
    L.rkfdOracleGetVolumePair.argtypes = [vp, C.c_int, vp, C.c_int]
    L.rkfdOracleVolumeLP.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp]
    if hasattr(L, "rkfdOracleRolloutsMT"):          # (not in the flop-counting build)
        L.rkfdOracleRolloutsMT.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_double, vp]; L.rkfdOracleRolloutsMT.restype = C.c_long
    return L


_libs = {}


def lib(path=None):
    """the oracle library (built on first use); `path`: another build of the same source (e.g. the fused-multiply-add
    build `make -C oracle fma`, used as a rounding control)"""
    global _lib
    if path is not None:
        if path not in _libs:
            _libs[path] = _bind(path)
        return _libs[path]
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = _bind(LIB_PATH)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One simulated world (= one rkFD of the reference) on the CPU."""

    def __init__(self, model_ptr, libpath=None):
        self._L = lib(libpath)
        self._model = model_ptr            # keep the owner alive
        m = model_ptr.contents
        self.ndof, self.nlink, self.ncand = m.ndof, m.nlink, m.ncand
        self._o = self._L.rkfdOracleCreate(C.cast(model_ptr, C.c_void_p))
        if not self._o:
            raise RuntimeError("the oracle does not cover this world (spherical / breakable-float joint)")

    def close(self):
        if getattr(self, "_o", None):
            self._L.rkfdOracleDestroy(self._o)
            self._o = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, dis, vel):
        dis = np.ascontiguousarray(dis, dtype=np.float64); vel = np.ascontiguousarray(vel, dtype=np.float64)
        assert dis.size == self.ndof and vel.size == self.ndof
        self._L.rkfdOracleSetState(self._o, _p(dis), _p(vel))

    def get_state(self):
        d = np.empty(self.ndof); v = np.empty(self.ndof); a = np.empty(self.ndof)
        self._L.rkfdOracleGetState(self._o, _p(d), _p(v), _p(a))
        return d, v, a

    def set_motor_input(self, inp):
        inp = np.ascontiguousarray(inp, dtype=np.float64)
        assert inp.size == self.nlink
        self._L.rkfdOracleSetMotorInput(self._o, _p(inp))

    def get_contact(self):
        act = np.empty(self.ncand, dtype=np.int32); typ = np.empty(self.ncand, dtype=np.int32)
        ref = np.empty((self.ncand, 3)); f = np.empty((self.ncand, 3))
        self._L.rkfdOracleGetContact(self._o, _p(act), _p(typ), _p(ref), _p(f))
        return act, typ, ref, f

    def set_contact(self, act, typ, ref):
        act = np.ascontiguousarray(act, dtype=np.int32); typ = np.ascontiguousarray(typ, dtype=np.int32)
        ref = np.ascontiguousarray(ref, dtype=np.float64)
        self._L.rkfdOracleSetContact(self._o, _p(act), _p(typ), _p(ref))

    def reset_contact(self):
        """forget all contact-vertex and friction-pivot state (a fresh world at the state set next)"""
        self.set_contact(np.zeros(self.ncand, dtype=np.int32), np.zeros(self.ncand, dtype=np.int32), np.zeros((self.ncand, 3)))
        self.set_pivot(np.zeros(self.nlink, dtype=np.int32), np.zeros(self.nlink))
        self.set_broken(np.zeros(self.nlink, dtype=np.int32))

    def get_pivot(self):
        typ = np.empty(self.nlink, dtype=np.int32); prev = np.empty(self.nlink)
        self._L.rkfdOracleGetPivot(self._o, _p(typ), _p(prev))
        return typ, prev

    def set_pivot(self, typ, prev):
        typ = np.ascontiguousarray(typ, dtype=np.int32); prev = np.ascontiguousarray(prev, dtype=np.float64)
        self._L.rkfdOracleSetPivot(self._o, _p(typ), _p(prev))

    def get_broken(self):
        b = np.empty(self.nlink, dtype=np.int32)
        self._L.rkfdOracleGetBroken(self._o, _p(b))
        return b

    def set_broken(self, broken):
        b = np.ascontiguousarray(broken, dtype=np.int32)
        self._L.rkfdOracleSetBroken(self._o, _p(b))

    def update_init(self):
        self._L.rkfdOracleUpdateInit(self._o)

    def update_n(self, nsteps):
        return self._L.rkfdOracleUpdateN(self._o, int(nsteps))

    def last_qp_iter(self):
        return self._L.rkfdOracleLastQPIter(self._o)

    def qp_cycle_stops(self):
        return self._L.rkfdOracleQPCycleStops(self._o)

    def volume_guard_hits(self):
        return self._L.rkfdOracleVolumeGuardHits(self._o)

    def update(self):
        return self._L.rkfdOracleUpdate(self._o)

    def eval(self, do_up_ref=False):
        return self._L.rkfdOracleEval(self._o, 1 if do_up_ref else 0)

    @property
    def time(self):
        return self._L.rkfdOracleTime(self._o)

    def link_frames(self):
        R = np.empty((self.nlink, 3, 3)); p = np.empty((self.nlink, 3))
        self._L.rkfdOracleGetLinkFrames(self._o, _p(R), _p(p))
        return R, p

    def link_vel_acc(self):
        v = np.empty((self.nlink, 6)); a = np.empty((self.nlink, 6))
        self._L.rkfdOracleGetLinkVelAcc(self._o, _p(v), _p(a))
        return v, a

    def volume_pairs(self):
        """Volume plugin: the colliding rigid pairs of the last evaluation, each a dict (pair, ntri, type, volume, center,
        norm, wrench (world, force 3 + torque 3 about the centre), q 6x6, c 6, planes [(v, n)])"""
        out = []
        for k in range(self._L.rkfdOracleVolumePairs(self._o)):
            buf = np.zeros(64 + 6 * 96)
            n = self._L.rkfdOracleGetVolumePair(self._o, k, _p(buf), buf.size)
            assert 0 < n <= buf.size
            ncp = int(buf[2])
            out.append(dict(pair=int(buf[0]), ntri=int(buf[1]), type=int(buf[3]), volume=buf[4], center=buf[5:8].copy(),
                            norm=buf[8:11].copy(), wrench=buf[11:17].copy(), q=buf[17:53].reshape(6, 6).copy(),
                            c=buf[53:59].copy(), planes=buf[59:59 + 6 * ncp].reshape(ncp, 6).copy()))
        return out

    def mlcp(self):
        cap = 3 * max(self.ncand, 1)
        a = np.zeros(cap * cap); b = np.zeros(cap); f = np.zeros(cap)
        nc = self._L.rkfdOracleGetMLCP(self._o, _p(a), _p(b), _p(f), cap)
        n3 = 3 * nc
        return nc, a[:n3 * n3].reshape(n3, n3).copy(), b[:n3].copy(), f[:n3].copy()


def volume_lp(A, b, c=None):
    """the oracle's simplex LP: min c'x s.t. Ax = b, x >= 0 (c None: any feasible vertex).  Returns x or None."""
    A = np.ascontiguousarray(A, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    mr, n = A.shape
    x = np.zeros(n)
    cc = None if c is None else np.ascontiguousarray(c, dtype=np.float64)
    ok = lib().rkfdOracleVolumeLP(mr, n, _p(A), _p(b), _p(cc), _p(x))
    return x if ok else None


def rollouts_mt(model_ptr, dis, vel, horizon, seconds, nthreads=1):
    """bench.py's CPU baseline, timed inside C: `nthreads` OS threads (each its own oracle) doing rollouts of `horizon` steps from
    the states dis / vel [ninst, ndof] for `seconds`; returns (steps done by all threads, wall seconds of the longest thread)"""
    dis = np.ascontiguousarray(dis, dtype=np.float64); vel = np.ascontiguousarray(vel, dtype=np.float64)
    el = C.c_double(0.0)
    n = lib().rkfdOracleRolloutsMT(C.cast(model_ptr, C.c_void_p), int(nthreads), int(dis.shape[0]), _p(dis), _p(vel), int(horizon), float(seconds), C.byref(el))
    if n < 0:
        raise RuntimeError("rkfdOracleRolloutsMT failed")
    return int(n), float(el.value)
