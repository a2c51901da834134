"""Known-answer tests of the oracle's restatement of the Vert plugin's rigid branch
(reference src/rkfd_vert.c:258-336, src/rkfd_opt_qp.c:43-181)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.pyoracle import lib


def _L():
    L = lib()
    vp = C.c_void_p
    L.rkfdOraclePinvSolve.argtypes = [C.c_int, vp, vp, vp]
    L.rkfdOracleQPASM.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    L.rkfdOracleQPASM.restype = C.c_int
    return L


def _pyramid(nc, P, mu):
    nf = np.zeros((P * nc, 3 * nc))
    for c in range(nc):
        for i in range(P):
            th = 2 * np.pi * i / P - np.pi / P
            nf[P * c + i, 3 * c:3 * c + 3] = (mu * np.cos(-np.pi / P), np.sin(th), np.cos(th))
    return nf


def test_moore_penrose_solve_matches_numpy_pinv():
    """zLESolveMP stand-in: symmetric indefinite and rank-deficient systems (a KKT matrix with the eight
    faces of one pyramid all active has rank 3 in that block)"""
    L = _L()
    rng = np.random.default_rng(5)
    for n, rank in ((6, 6), (12, 7), (30, 30), (30, 19), (44, 25)):
        B = rng.normal(size=(n, rank)); s = rng.choice([-1.0, 1.0], size=rank) * rng.uniform(0.5, 3.0, rank)
        K = np.ascontiguousarray((B * s) @ B.T)
        rhs = rng.normal(size=n); x = np.empty(n)
        L.rkfdOraclePinvSolve(n, K.ctypes.data, rhs.ctypes.data, x.ctypes.data)
        ref = np.linalg.pinv(K, rcond=1e-12) @ rhs
        assert np.abs(x - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), (n, rank)


@pytest.mark.parametrize("seed", range(12))
def test_active_set_qp_satisfies_kkt(seed):
    """the returned point is THE minimiser of the strictly convex QP: feasible, stationary with
    non-negative multipliers on the returned active set, complementary (solver-independent check)"""
    L = _L()
    rng = np.random.default_rng(seed)
    nc = int(rng.integers(1, 6)); P = 8; n = 3 * nc; mc = P * nc
    A = rng.normal(size=(n, n)); q = np.ascontiguousarray(A.T @ A + 1e-2 * np.eye(n))
    c = rng.normal(size=n) * 3.0
    nf = np.ascontiguousarray(_pyramid(nc, P, 0.5)); d = np.zeros(mc)
    ans = np.empty(n); idx = np.zeros(mc, dtype=np.int32)
    it = L.rkfdOracleQPASM(n, mc, P, q.ctypes.data, c.ctypes.data, nf.ctypes.data, d.ctypes.data, ans.ctypes.data, idx.ctypes.data)
    assert 1 <= it < 200
    g = nf @ ans
    assert g.min() > -1e-9                                   # primal feasibility
    assert np.all(np.abs(g[idx != 0]) < 1e-9)                # the active set is active
    grad = q @ ans + c                                        # stationarity: grad = nf' y, y >= 0 on the active set
    Na = nf[idx != 0]
    if Na.shape[0] == 0:
        assert np.abs(grad).max() < 1e-8
    else:
        from scipy.optimize import nnls
        y, res = nnls(Na.T, grad)
        assert res < 1e-7 * max(1.0, np.abs(grad).max())
    # and it is no worse than a brute-force projected-gradient reference
    x = ans.copy()
    f0 = 0.5 * ans @ q @ ans + c @ ans
    for _ in range(50):
        trial = x + rng.normal(size=n) * 1e-3
        if (nf @ trial).min() >= 0:
            assert 0.5 * trial @ q @ trial + c @ trial >= f0 - 1e-12


def test_vert_rigid_box_comes_to_rest(R, oracle_cls):
    """box.ztk on floor.ztk under the Vert plugin with the RIGID 'ground body' entry: the four bottom
    vertices end up in static friction, the box neither sinks nor flies off, and the time-average of the
    normal force over the scheme's limit cycle carries the weight"""
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VERT); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "box.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    dis = np.zeros(6); dis[2] = 0.0499; vel = np.zeros(6); vel[0] = 0.05
    o = oracle_cls(w.model); o.set_state(dis, vel); o.update_init(); o.update_n(300)
    fz, zs = [], []
    for _ in range(40):
        o.update()
        act, typ, ref, f = o.get_contact(); d, v, a = o.get_state()
        fz.append(f[:, 2].sum()); zs.append(d[2])
        assert act.sum() == 4 and (typ[act != 0] == R.SF).all()
    assert abs(np.mean(zs) - 0.05) < 1e-4 and np.ptp(zs) < 1e-4
    assert 0.5 * 0.5 * 9.80665 < np.mean(fz) < 2.0 * 0.5 * 9.80665
