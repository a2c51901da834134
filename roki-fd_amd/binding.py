"""ctypes binding of librkfd_amd.so (include/rkfd_hip.h, include/roki_fd_amd.h)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librkfd_amd.so")

JOINT_FIXED, JOINT_REVOL, JOINT_PRISM, JOINT_FLOAT = 0, 1, 2, 3
SOLVER_VERT, SOLVER_MLCP, SOLVER_VOLUME = 0, 1, 2
CONTACT_RIGID, CONTACT_ELASTIC = 0, 1
SF, KF = 0, 1

_pi = C.POINTER(C.c_int)
_pd = C.POINTER(C.c_double)


class RkfdError(RuntimeError):
    pass


class RkfdModel(C.Structure):
    """Mirror of rkfdModel (include/rkfd_model.h); field order must match."""
    _fields_ = [
        ("nlink", C.c_int), ("ndof", C.c_int), ("nchain", C.c_int),
        ("parent", _pi), ("jtype", _pi), ("dofoff", _pi), ("chain", _pi),
        ("org", _pd), ("mass", _pd), ("com", _pd), ("inertia", _pd),
        ("stiff", _pd), ("visc", _pd), ("coulomb", _pd), ("sfric", _pd),
        ("mtype", _pi),
        ("mot_k", _pd), ("mot_admit", _pd), ("mot_vmax", _pd), ("mot_vmin", _pd), ("mot_gear", _pd), ("mot_inertia", _pd),
        ("nshape", C.c_int),
        ("shape_link", _pi), ("shape_voff", _pi), ("shape_foff", _pi),
        ("verts", _pd), ("planes", _pd),
        ("shape_slide_mode", _pi), ("shape_slide_vel", _pd), ("shape_slide_axis", _pd),
        ("npair", C.c_int),
        ("pair_shape", _pi), ("pair_ci", _pi),
        ("nci", C.c_int),
        ("ci_type", _pi),
        ("ci_sf", _pd), ("ci_kf", _pd), ("ci_k", _pd), ("ci_l", _pd), ("ci_e", _pd), ("ci_v", _pd),
        ("ncand", C.c_int),
        ("cand_pair", _pi), ("cand_side", _pi), ("cand_vert", _pi),
        ("dt", C.c_double), ("friction_weight", C.c_double),
        ("max_iter", C.c_int), ("solver", C.c_int), ("pyramid", C.c_int),
        ("brk_f", _pd), ("brk_t", _pd),
    ]

    def arr(self, name, n, dtype=None):
        p = getattr(self, name)
        if n == 0:
            return np.zeros(0, dtype=dtype or (np.int32 if isinstance(p, _pi) else np.float64))
        return np.ctypeslib.as_array(p, shape=(n,)).copy()


_lib = None


def lib():
    """Loads librkfd_amd.so; raises when it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RkfdError(f"{LIB_PATH} is missing: run `make` (or __graft_entry__.build()) first")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rkfdWorldCreate.restype = vp
    L.rkfdWorldFree.argtypes = [vp]
    L.rkfdWorldRegFile.argtypes = [vp, C.c_char_p]
    L.rkfdWorldSetContactInfo.argtypes = [vp, C.c_char_p]
    L.rkfdWorldPairChainUnreg.argtypes = [vp, C.c_int]
    L.rkfdWorldSetPrp.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.c_int]
    L.rkfdWorldSetPyramid.argtypes = [vp, C.c_int]
    L.rkfdWorldSetSlide.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, vp]; L.rkfdWorldSetSlide.restype = C.c_int
    L.rkfdWorldModel.argtypes = [vp]
    L.rkfdWorldModel.restype = C.POINTER(RkfdModel)
    L.rkfdWorldChainDofOffset.argtypes = [vp, C.c_int]
    L.rkfdWorldChainLinkOffset.argtypes = [vp, C.c_int]
    L.rkfdWorldChainInitDis.argtypes = [vp, C.c_int, _pd]
    L.rkfdWorldWriteZTK.argtypes = [vp, C.c_int, C.c_char_p, _pd]
    L.rkfdHipLastError.restype = C.c_char_p
    L.rkfdBatchCreate.argtypes = [C.POINTER(RkfdModel), C.c_int, C.c_int, C.c_int]
    L.rkfdBatchCreate.restype = vp
    L.rkfdBatchDestroy.argtypes = [vp]
    L.rkfdSpecializeCompile.argtypes = [C.POINTER(RkfdModel), C.c_int]
    for f in ("rkfdBatchSize", "rkfdBatchDof", "rkfdBatchLdsBytes", "rkfdBatchResidency", "rkfdBatchSpecialize"):
        getattr(L, f).argtypes = [vp]
    L.rkfdBatchSetState.argtypes = [vp, vp, vp]
    L.rkfdBatchGetState.argtypes = [vp, vp, vp, vp]
    L.rkfdBatchSetMotorInput.argtypes = [vp, vp]
    L.rkfdBatchGetContact.argtypes = [vp, vp, vp, vp, vp]
    L.rkfdBatchSetContact.argtypes = [vp, vp, vp, vp]
    L.rkfdBatchGetPivot.argtypes = [vp, vp, vp]
    L.rkfdBatchSetPivot.argtypes = [vp, vp, vp]
    L.rkfdBatchGetBroken.argtypes = [vp, vp]; L.rkfdBatchSetBroken.argtypes = [vp, vp]
    L.rkfdBatchSetInstancesPerWave.argtypes = [vp, C.c_int]; L.rkfdBatchInstancesPerWave.argtypes = [vp]
    L.rkfdBatchSetStepsPerLaunch.argtypes = [vp, C.c_int]
    L.rkfdBatchTuneInstancesPerWave.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.rkfdSpecializeCompileW.argtypes = [C.POINTER(RkfdModel), C.c_int, C.c_int]
    L.rkfdBatchUpdateInit.argtypes = [vp, vp]
    L.rkfdBatchUpdate.argtypes = [vp, C.c_int, vp]
    L.rkfdBatchEval.argtypes = [vp, C.c_int, vp]
    L.rkfdLdsBytesFor.argtypes = [C.POINTER(RkfdModel), C.c_int]
    L.rkfdBatchStatus.argtypes = [vp, vp]
    L.rkfdBatchContactStats.argtypes = [vp, C.c_int, _pd, _pd, C.POINTER(C.c_longlong)]
    L.rkfdBatchSnapshot.argtypes = [vp]; L.rkfdBatchRestore.argtypes = [vp, vp]
    L.rkfdBatchProfile.argtypes = [vp, C.c_int, vp]
    L.rkfdBatchSetSplit.argtypes = [vp, C.c_int]; L.rkfdBatchJoin.argtypes = [vp, vp]
    L.rkfdBatchTimeLaunches.argtypes = [vp, C.c_int]; L.rkfdBatchLaunchTiming.argtypes = [vp, vp, vp]
    for f in ("rkfdBatchDevDis", "rkfdBatchDevVel", "rkfdBatchDevAcc"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = vp
    L.rkfdNodeCreate.argtypes = [C.POINTER(RkfdModel), C.c_int, C.c_int, C.c_int, vp]; L.rkfdNodeCreate.restype = vp
    L.rkfdNodeDestroy.argtypes = [vp]
    for f in ("rkfdNodeDevices", "rkfdNodeSize", "rkfdNodeSpecialize", "rkfdNodeUpdateInit", "rkfdNodeSnapshot", "rkfdNodeRestore", "rkfdNodeStatus"):
        getattr(L, f).argtypes = [vp]
    L.rkfdNodeShard.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rkfdNodeBatch.argtypes = [vp, C.c_int]; L.rkfdNodeBatch.restype = vp
    L.rkfdNodeSetState.argtypes = [vp, vp, vp]; L.rkfdNodeSetMotorInput.argtypes = [vp, vp]
    L.rkfdNodeGetState.argtypes = [vp, vp, vp, vp]
    L.rkfdNodeSetSplit.argtypes = [vp, C.c_int]; L.rkfdNodeUpdate.argtypes = [vp, C.c_int]
    L.rkfdNodeSetStepsPerLaunch.argtypes = [vp, C.c_int]; L.rkfdNodeTuneInstancesPerWave.argtypes = [vp, C.c_int]
    L.rkfdNodeGather.argtypes = [vp, vp, vp]
    L.rkfdNodeGatherDev.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]; L.rkfdNodeGatherDev.restype = vp
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class World:
    """The rkFD world builder: chains registered from ZTK files + contact-info table.
    Mirrors rkFDCreate / rkFDChainRegFile / rkFDContactInfoScanFile / rkFDSetSolver /
    rkFDPrpSet* (reference src/rkfd_sim.c:32-54,224-273; include/roki_fd/rkfd_sim.h:89-93)."""

    def __init__(self, solver=SOLVER_VERT, dt=0.001, friction_weight=100.0, max_iter=10):
        self._L = lib()
        self._w = self._L.rkfdWorldCreate()
        if not self._w:
            raise RkfdError("rkfdWorldCreate failed")
        self.nchain = 0
        self.set_prp(dt, friction_weight, max_iter, solver)

    def close(self):
        if self._w:
            self._L.rkfdWorldFree(self._w)
            self._w = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_slide(self, chain, shape, mode, vel=0.0, axis=(0.0, 1.0, 0.0)):
        """rkFDCDCellSetSlideMode / Vel / Axis on shape number `shape` of a chain (its order in the ZTK file)"""
        ax = (C.c_double * 3)(*axis)
        if self._L.rkfdWorldSetSlide(self._w, int(chain), int(shape), int(bool(mode)), float(vel), ax) != 0:
            raise RkfdError("no such chain / shape")

    def set_pyramid(self, pyramid):
        """rkFDPrpSetPyramid: faces of the Vert plugin's friction pyramid (default 8)"""
        self._L.rkfdWorldSetPyramid(self._w, int(pyramid))

    def set_prp(self, dt, friction_weight, max_iter, solver):
        self._L.rkfdWorldSetPrp(self._w, dt, friction_weight, max_iter, solver)

    def reg_file(self, path):
        cid = self._L.rkfdWorldRegFile(self._w, os.fspath(path).encode())
        if cid < 0:
            raise RkfdError(f"cannot register chain from {path}")
        self.nchain = cid + 1
        return cid

    def contact_info(self, path):
        if self._L.rkfdWorldSetContactInfo(self._w, os.fspath(path).encode()) != 0:
            raise RkfdError(f"cannot read contact info from {path}")

    def pair_chain_unreg(self, chain):
        self._L.rkfdWorldPairChainUnreg(self._w, chain)

    @property
    def model(self):
        p = self._L.rkfdWorldModel(self._w)
        if not p:
            raise RkfdError("rkfdWorldModel failed")
        return p

    def dof_offset(self, chain):
        return self._L.rkfdWorldChainDofOffset(self._w, chain)

    def link_offset(self, chain):
        return self._L.rkfdWorldChainLinkOffset(self._w, chain)

    def init_dis(self, chain):
        buf = np.zeros(64, dtype=np.float64)
        n = self._L.rkfdWorldChainInitDis(self._w, chain, buf.ctypes.data_as(_pd))
        return buf[:n].copy()


class Batch:
    """B instances of one world on one GPU (include/rkfd_hip.h).  All arrays are
    instance-major numpy arrays [B, ...]."""

    def __init__(self, world, batch, device=0, max_rigid=8):
        self._L = lib()
        self.world = world
        m = world.model.contents
        self.B, self.ndof, self.nlink, self.ncand = batch, m.ndof, m.nlink, m.ncand
        self._b = self._L.rkfdBatchCreate(world.model, batch, device, max_rigid)
        if not self._b:
            raise RkfdError(self._L.rkfdHipLastError().decode())

    def close(self):
        if getattr(self, "_b", None):
            self._L.rkfdBatchDestroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r < 0:
            raise RkfdError(self._L.rkfdHipLastError().decode())

    def set_state(self, dis, vel):
        dis = np.ascontiguousarray(dis, dtype=np.float64).reshape(self.B, self.ndof)
        vel = np.ascontiguousarray(vel, dtype=np.float64).reshape(self.B, self.ndof)
        self._chk(self._L.rkfdBatchSetState(self._b, _ptr(dis), _ptr(vel)))

    def get_state(self):
        dis = np.empty((self.B, self.ndof)); vel = np.empty_like(dis); acc = np.empty_like(dis)
        self._chk(self._L.rkfdBatchGetState(self._b, _ptr(dis), _ptr(vel), _ptr(acc)))
        return dis, vel, acc

    def set_motor_input(self, inp):
        inp = np.ascontiguousarray(inp, dtype=np.float64).reshape(self.B, self.nlink)
        self._chk(self._L.rkfdBatchSetMotorInput(self._b, _ptr(inp)))

    def get_contact(self):
        act = np.empty((self.B, self.ncand), dtype=np.int32); typ = np.empty_like(act)
        ref = np.empty((self.B, self.ncand, 3)); f = np.empty_like(ref)
        self._chk(self._L.rkfdBatchGetContact(self._b, _ptr(act), _ptr(typ), _ptr(ref), _ptr(f)))
        return act, typ, ref, f

    def set_contact(self, act, typ, ref):
        act = np.ascontiguousarray(act, dtype=np.int32); typ = np.ascontiguousarray(typ, dtype=np.int32)
        ref = np.ascontiguousarray(ref, dtype=np.float64)
        self._chk(self._L.rkfdBatchSetContact(self._b, _ptr(act), _ptr(typ), _ptr(ref)))

    def get_pivot(self):
        typ = np.empty((self.B, self.nlink), dtype=np.int32); prev = np.empty((self.B, self.nlink))
        self._chk(self._L.rkfdBatchGetPivot(self._b, _ptr(typ), _ptr(prev)))
        return typ, prev

    def set_pivot(self, typ, prev):
        typ = np.ascontiguousarray(typ, dtype=np.int32); prev = np.ascontiguousarray(prev, dtype=np.float64)
        self._chk(self._L.rkfdBatchSetPivot(self._b, _ptr(typ), _ptr(prev)))

    def get_broken(self):
        """breakable float joints: 1 per link whose joint has broken, [B, nlink]"""
        br = np.empty((self.B, self.nlink), dtype=np.int32)
        self._chk(self._L.rkfdBatchGetBroken(self._b, _ptr(br)))
        return br

    def set_broken(self, broken):
        br = np.ascontiguousarray(broken, dtype=np.int32).reshape(self.B, self.nlink)
        self._chk(self._L.rkfdBatchSetBroken(self._b, _ptr(br)))

    def update_init(self, stream=None):
        self._chk(self._L.rkfdBatchUpdateInit(self._b, stream))

    def update(self, nsteps=1, stream=None):
        self._chk(self._L.rkfdBatchUpdate(self._b, nsteps, stream))

    def eval(self, do_up_ref=False, stream=None):
        self._chk(self._L.rkfdBatchEval(self._b, 1 if do_up_ref else 0, stream))

    def status(self, stream=None):
        r = self._L.rkfdBatchStatus(self._b, stream)
        if r < 0:
            raise RkfdError(self._L.rkfdHipLastError().decode())
        return r

    def contact_stats(self, reset=False):
        """(mean rigid, mean elastic contact vertices per instance-step, instance-steps counted) since the last reset"""
        rg = C.c_double(); el = C.c_double(); n = C.c_longlong()
        self._chk(self._L.rkfdBatchContactStats(self._b, int(bool(reset)), C.byref(rg), C.byref(el), C.byref(n)))
        return rg.value, el.value, n.value

    def snapshot(self):
        """keep a device-resident copy of the whole state (start of MPC-style rollouts)"""
        self._chk(self._L.rkfdBatchSnapshot(self._b))

    def restore(self, stream=None):
        """put the snapshot back, in stream order (no host traffic)"""
        self._chk(self._L.rkfdBatchRestore(self._b, C.c_void_p(stream or 0)))

    def set_split(self, nsplit):
        """rkfdBatchSetSplit: launch the batch as nsplit kernels on internal streams (tails overlap)"""
        self._chk(self._L.rkfdBatchSetSplit(self._b, int(nsplit)))

    def join(self, stream=None):
        """rkfdBatchJoin: make `stream` wait for the split launches (no host synchronisation)"""
        self._chk(self._L.rkfdBatchJoin(self._b, C.c_void_p(stream or 0)))

    def time_launches(self, on=True):
        self._chk(self._L.rkfdBatchTimeLaunches(self._b, int(bool(on))))

    def launch_timing(self):
        """(number of launches, their summed duration in ms) since time_launches(True); synchronises the device"""
        n = C.c_int(); ms = C.c_double()
        self._chk(self._L.rkfdBatchLaunchTiming(self._b, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile(self, nsteps=1):
        """diagnostic launch with in-kernel phase stamps: [B, 32] cycles (RKFD_NPROF)"""
        out = np.zeros((self.B, 32), dtype=np.uint64)
        self._chk(self._L.rkfdBatchProfile(self._b, nsteps, _ptr(out)))
        return out

    @property
    def lds_bytes(self):
        return self._L.rkfdBatchLdsBytes(self._b)

    def set_steps_per_launch(self, n):
        """under split launches: steps one launch carries (default 5)"""
        self._chk(self._L.rkfdBatchSetStepsPerLaunch(self._b, int(n)))

    def set_instances_per_wave(self, ipw):
        """1 (default) or 2 instances per wavefront in the world-specific kernel; call before specialize()"""
        self._chk(self._L.rkfdBatchSetInstancesPerWave(self._b, int(ipw)))

    def instances_per_wave(self):
        return self._L.rkfdBatchInstancesPerWave(self._b)

    def tune_instances_per_wave(self, nsteps=8):
        """measure both mappings on the present state (kept), keep the faster; -> (chosen, (ms with 1, ms with 2))"""
        ms = (C.c_double * 2)()
        r = self._L.rkfdBatchTuneInstancesPerWave(self._b, int(nsteps), ms)
        self._chk(r)
        return r, (ms[0], ms[1])

    def specialize(self):
        """compile the step kernel for this world (hipRTC): same results, its dimensions as literals"""
        self._chk(self._L.rkfdBatchSpecialize(self._b))

    def residency(self):
        """instances per compute unit the HIP runtime can keep resident (registers + LDS)"""
        return self._L.rkfdBatchResidency(self._b)

    def dev_tensors(self):
        """torch tensors ALIASING the live device state [B, ndof] (dis, vel, acc): zero-copy views
        for consumers on the device, e.g. the RCCL all-gather of final states."""
        import torch

        class _View:
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = dict(shape=shape, typestr="<f8", data=(int(ptr), False), version=2, strides=None)
        return tuple(torch.as_tensor(_View(p, (self.B, self.ndof)), device="cuda") for p in self.dev_ptrs())

    def dev_ptrs(self):
        return (self._L.rkfdBatchDevDis(self._b), self._L.rkfdBatchDevVel(self._b), self._L.rkfdBatchDevAcc(self._b))


class Node:
    """`total` instances of one world over the GPUs of one node from ONE process (include/rkfd_hip.h: rkfdNode*): device k
    simulates its contiguous shard with its own host thread and stream inside the library, no per-step communication; the
    only collective is gather(): one RCCL all-gather of the final {dis, vel}."""

    def __init__(self, world, total, max_rigid=8, ndev=0, devices=None):
        self._L = lib()
        self.world = world
        m = world.model.contents
        self.total, self.ndof, self.nlink = total, m.ndof, m.nlink
        dv = None
        if devices is not None:
            dv = (C.c_int * len(devices))(*devices); ndev = len(devices)
        self._n = self._L.rkfdNodeCreate(world.model, total, max_rigid, ndev, C.cast(dv, C.c_void_p) if dv is not None else None)
        if not self._n:
            raise RkfdError(self._L.rkfdHipLastError().decode())
        self.ndev = self._L.rkfdNodeDevices(self._n)

    def close(self):
        if getattr(self, "_n", None):
            self._L.rkfdNodeDestroy(self._n)
            self._n = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r < 0:
            raise RkfdError(self._L.rkfdHipLastError().decode())
        return r

    def shards(self):
        out = []
        for k in range(self.ndev):
            d, lo, hi = C.c_int(), C.c_int(), C.c_int()
            self._chk(self._L.rkfdNodeShard(self._n, k, C.byref(d), C.byref(lo), C.byref(hi)))
            out.append((d.value, lo.value, hi.value))
        return out

    def set_state(self, dis, vel):
        dis = np.ascontiguousarray(dis, dtype=np.float64).reshape(self.total, self.ndof)
        vel = np.ascontiguousarray(vel, dtype=np.float64).reshape(self.total, self.ndof)
        self._chk(self._L.rkfdNodeSetState(self._n, _ptr(dis), _ptr(vel)))

    def set_motor_input(self, inp):
        inp = np.ascontiguousarray(inp, dtype=np.float64).reshape(self.total, self.nlink)
        self._chk(self._L.rkfdNodeSetMotorInput(self._n, _ptr(inp)))

    def get_state(self):
        dis = np.empty((self.total, self.ndof)); vel = np.empty_like(dis); acc = np.empty_like(dis)
        self._chk(self._L.rkfdNodeGetState(self._n, _ptr(dis), _ptr(vel), _ptr(acc)))
        return dis, vel, acc

    def specialize(self):
        self._chk(self._L.rkfdNodeSpecialize(self._n))

    def set_split(self, nsplit):
        self._chk(self._L.rkfdNodeSetSplit(self._n, nsplit))

    def set_steps_per_launch(self, n):
        self._chk(self._L.rkfdNodeSetStepsPerLaunch(self._n, int(n)))

    def tune_instances_per_wave(self, nsteps=8):
        self._chk(self._L.rkfdNodeTuneInstancesPerWave(self._n, int(nsteps)))

    def update_init(self):
        self._chk(self._L.rkfdNodeUpdateInit(self._n))

    def update(self, nsteps=1):
        self._chk(self._L.rkfdNodeUpdate(self._n, nsteps))

    def snapshot(self):
        self._chk(self._L.rkfdNodeSnapshot(self._n))

    def restore(self):
        self._chk(self._L.rkfdNodeRestore(self._n))

    def status(self):
        return self._chk(self._L.rkfdNodeStatus(self._n))

    def gather(self):
        """one RCCL all-gather of the final {dis, vel}; returns them on the host, [total, ndof] each"""
        dis = np.empty((self.total, self.ndof)); vel = np.empty_like(dis)
        self._chk(self._L.rkfdNodeGather(self._n, _ptr(dis), _ptr(vel)))
        return dis, vel
