"""Known-answer tests of the oracle's Volume plugin (reference src/rkfd_volume.c; oracle/rkfd_oracle_volume.h).
The reference ships no vectors for it either: these pin the restatement to mechanics."""
import os

import numpy as np
import pytest

G = 9.80665


def _box_world(R, second=None, floor="floor.ztk"):
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VOLUME)
    w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "box.ztk"))
    if second:
        w.reg_file(os.path.join(M, second))
    w.reg_file(os.path.join(M, floor))
    return w


def test_intersection_volume_of_a_box_sunk_into_the_floor(R, oracle_cls):
    """depth d: volume = area d, centre d/2 under the floor, normal up; Q = integral of [1 -[p x]; [p x] -[p x]^2] over the
    top and bottom faces (symmetric, positive semi-definite), c = -K V (n; 0); the contact polygon is the square"""
    w = _box_world(R)
    o = oracle_cls(w.model)
    for d in (1e-4, 2.5e-3):
        dis = np.zeros(6); dis[:3] = (0.2, -0.1, 0.05 - d); dis[5] = 0.3
        o.set_state(dis, np.zeros(6)); assert o.eval(False) == 0
        (p,) = o.volume_pairs()
        assert np.isclose(p["volume"], 0.01 * d, rtol=1e-9)
        assert np.allclose(p["center"], [0.2, -0.1, -d / 2], atol=1e-12)
        assert np.allclose(p["norm"], [0, 0, 1], atol=1e-12)
        q = p["q"]
        assert np.allclose(q, q.T, atol=1e-15) and np.linalg.eigvalsh(q).min() > -1e-15
        assert np.allclose(q[:3, :3], 2 * 0.01 * np.eye(3), atol=1e-12)
        # second moments of a 0.1 square about its centre, twice: Ixx = Iyy = a^4/12, Izz = a^4/6
        assert np.allclose(np.diag(q)[3:], 2 * np.array([1e-4 / 12, 1e-4 / 12, 1e-4 / 6]), rtol=1e-9)
        assert np.allclose(p["c"], [0, 0, -1000.0 * 0.01 * d, 0, 0, 0], atol=1e-12)
        assert len(p["planes"]) == 4
        for v, n in zip(p["planes"][:, :3], p["planes"][:, 3:]):
            assert abs(np.linalg.norm(n) - 1) < 1e-12 and abs(n[2]) < 1e-12
            assert np.isclose(-n @ v, 0.05, atol=1e-12)            # the centre lies 0.05 inside every edge


def test_box_comes_to_rest_carrying_its_weight(R, oracle_cls):
    """dropped tilted, the box ends flat: the pair's wrench is m g along the normal, no moment about the centre of the
    contact area, static friction"""
    w = _box_world(R)
    o = oracle_cls(w.model)
    sc = R.scenarios.config1_rigid(batch=1)
    dis = sc["dis"][0].copy(); dis[:3] = (0, 0, 0.1)
    o.set_state(dis, np.zeros(6)); o.update_init()
    for _ in range(2000):
        assert o.update() == 0
    d, v, a = o.get_state()
    assert np.abs(v).max() < 1e-8 and abs(d[2] - 0.05) < 1e-6
    (p,) = o.volume_pairs()
    assert np.allclose(p["wrench"][:3], [0, 0, 0.5 * G], atol=1e-7) and np.abs(p["wrench"][3:]).max() < 1e-6
    assert p["type"] == R.SF


def test_sliding_box_is_braked_by_kinetic_friction(R, oracle_cls):
    """every evaluation in sliding contact: tangential force = weight(v) kf fn against the motion (weight = 1 at this speed);
    mean deceleration kf g"""
    w = _box_world(R)
    o = oracle_cls(w.model)
    dis = np.zeros(6); dis[2] = 0.04999
    vel = np.zeros(6); vel[0] = 0.5
    o.set_state(dis, vel); o.update_init()
    seen = 0
    for k in range(100):
        assert o.update() == 0
        ps = o.volume_pairs()
        if ps and ps[0]["wrench"][2] > 0:
            vx = o.get_state()[1][0]
            f, nrm = ps[0]["wrench"][:3], ps[0]["norm"]
            fn = f @ nrm; ft = np.linalg.norm(f - fn * nrm)
            # every vertex of the contact polygon slides its own way (the box yaws a little): the resultant is at most kf fn
            assert 0.29 < ft / fn <= 0.3 * (1 + 1e-9) and f[0] < 0 and ps[0]["type"] == R.KF
            seen += 1
    assert seen > 50
    assert np.isclose((0.5 - o.get_state()[1][0]) / 0.1, 0.3 * G, rtol=0.05)      # (the box bounces a little: the mean normal force is m g within a few per cent)


def test_two_moving_bodies_exchange_opposite_wrenches(R, oracle_cls):
    """a small box on the box on the floor: with both pairs in contact the momentum of the two boxes changes by gravity and
    the floor's wrench only (the inner pair's wrench cancels)"""
    w = _box_world(R, second="box_small.ztk")
    o = oracle_cls(w.model)
    m = w.model.contents
    assert m.ndof == 12
    hs = None
    # heights: lower box centre 0.05 - d, the small one rests on top of it
    import ctypes
    vz = np.array([m.verts[3 * i + 2] for i in range(m.shape_voff[1], m.shape_voff[2])])
    hs = vz.max()
    dis = np.zeros(12); dis[2] = 0.05 - 1e-4; dis[8] = 0.1 - 1e-4 + hs - 1e-4
    o.set_state(dis, np.zeros(12)); assert o.eval(True) == 0
    ps = {p["pair"]: p for p in o.volume_pairs()}
    assert len(ps) == 2
    acc = o.get_state()[2]
    mass = np.array([m.mass[i] for i in range(m.nlink)])
    lower, upper = mass[0], mass[1]
    # floor pair pushes the lower box up; the box-box pair acts on both
    tot = lower * acc[2] + upper * acc[8]
    ffloor = [p for p in ps.values() if abs(p["center"][2]) < 1e-3][0]["wrench"][2]
    assert np.isclose(tot, ffloor - (lower + upper) * G, rtol=1e-9, atol=1e-9)


def test_simplex_lp_against_scipy(oracle_cls):
    """the restated zLPSolveSimplex / zLPFeasibleBase: optimal values equal scipy's HiGHS on random feasible problems;
    infeasible problems are reported"""
    from scipy.optimize import linprog
    from oracle.pyoracle import volume_lp
    rng = np.random.default_rng(7)
    for trial in range(40):
        mr, n = int(rng.integers(1, 7)), int(rng.integers(6, 30))
        A = rng.normal(size=(mr, n)); x0 = rng.uniform(0, 1, n) * (rng.uniform(size=n) < 0.5)
        b = A @ x0
        c = rng.uniform(0.1, 1.0, n)            # positive cost: bounded
        x = volume_lp(A, b, c)
        ref = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs")
        assert x is not None and ref.status == 0
        assert np.allclose(A @ x, b, atol=1e-9) and x.min() > -1e-12
        assert np.isclose(c @ x, ref.fun, rtol=1e-8, atol=1e-10)
        xf = volume_lp(A, b)
        assert xf is not None and np.allclose(A @ xf, b, atol=1e-9) and xf.min() > -1e-12
    # infeasible: a positive combination cannot give a negative sum
    A = np.vstack([np.ones(5), rng.normal(size=(2, 5))])
    assert volume_lp(A, np.array([-1.0, 0.0, 0.0])) is None
