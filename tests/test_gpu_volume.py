"""GPU tier: the Volume plugin's device path (csrc/device/rkfd_dev_volume.h; reference src/rkfd_volume.c) through the C ABI
against the oracle's restatement (oracle/rkfd_oracle_volume.h)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _world(R, second=None, floor="floor.ztk"):
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VOLUME)
    w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "box.ztk"))
    if second:
        w.reg_file(os.path.join(M, second))
    w.reg_file(os.path.join(M, floor))
    return w


def _states(n, B, seed):
    """B different starts: flat just inside the floor / sliding / tilted above it"""
    rng = np.random.default_rng(seed)
    dis = np.zeros((B, n)); vel = np.zeros((B, n))
    for b in range(B):
        dis[b, :2] = rng.uniform(-0.2, 0.2, 2)
        if b % 3 == 0:
            dis[b, 2] = 0.05 - rng.uniform(1e-5, 2e-4)
        elif b % 3 == 1:
            dis[b, 2] = 0.05 - rng.uniform(1e-5, 2e-4); vel[b, :2] = rng.uniform(-0.6, 0.6, 2); vel[b, 5] = rng.uniform(-2, 2)
        else:
            dis[b, 2] = 0.075; dis[b, 3:6] = rng.uniform(-0.5, 0.5, 3); vel[b, :3] = rng.uniform(-0.3, 0.3, 3)
    return dis, vel


def test_volume_steps_match_oracle_from_the_oracles_states(R, oracle_cls):
    """400 steps of 12 boxes (resting, sliding, dropped tilted: face, edge and corner contacts, static and kinetic
    friction), every step started from the oracle's state.  Typical deviation 1e-13; a corner contact with an intersection volume
    of 1e-10 m^3 reaches 1e-7 (see tests/test_emu_parity.py::test_emulated_volume_plugin_matches_oracle)."""
    w = _world(R)
    B, n = 12, 6
    dis, vel = _states(n, B, 11)
    bt = R.Batch(w, B, max_rigid=4)
    os_ = []
    for b in range(B):
        o = oracle_cls(w.model); o.set_state(dis[b], vel[b]); o.update_init(); os_.append(o)
    errs = []; contact_steps = 0; kinds = set()
    for k in range(400):
        sd = np.array([o.get_state()[0] for o in os_]); sv = np.array([o.get_state()[1] for o in os_])
        bt.set_state(sd, sv); bt.update_init(); bt.update(1)
        assert bt.status() == 0, R.last_error()
        d, v, a = bt.get_state()
        for b, o in enumerate(os_):
            assert o.update() == 0
            od, ov, oa = o.get_state()
            errs.append(max(np.abs(d[b] - od).max(), np.abs(v[b] - ov).max(), np.abs(a[b] - oa).max() / max(1.0, np.abs(oa).max())))
            ps = o.volume_pairs()
            if ps and np.abs(ps[0]["wrench"]).max() > 0:
                contact_steps += 1; kinds.add((len(ps[0]["planes"]), ps[0]["type"]))
    errs = np.array(errs)
    assert contact_steps > 1500 and len(kinds) >= 4, (contact_steps, kinds)
    assert np.median(errs) < 1e-11 and np.quantile(errs, 0.99) < 1e-8 and errs.max() < 1e-5, (np.median(errs), np.quantile(errs, 0.99), errs.max())


def test_volume_free_running_trajectories(R, oracle_cls):
    """without re-synchronisation: 300 steps; boxes landing flat stay within 1e-8 of the oracle, the tumbling ones within 1e-4
    (every bounce amplifies the last digits; the oracle built with fused multiply-adds drifts the same way)"""
    w = _world(R)
    B, n = 9, 6
    dis, vel = _states(n, B, 5)
    bt = R.Batch(w, B, max_rigid=4)
    bt.set_state(dis, vel); bt.update_init(); bt.update(300)
    assert bt.status() == 0, R.last_error()
    d, v, a = bt.get_state()
    for b in range(B):
        o = oracle_cls(w.model); o.set_state(dis[b], vel[b]); o.update_init(); o.update_n(300)
        od, ov, oa = o.get_state()
        tol = 1e-8 if b % 3 == 0 else 1e-4
        assert np.abs(d[b] - od).max() < tol and np.abs(v[b] - ov).max() < tol * 100, (b, np.abs(d[b] - od).max(), np.abs(v[b] - ov).max())


def test_volume_three_pairs_two_moving_bodies(R, oracle_cls):
    """a small box on the box on the floor: three rigid pairs, one of them between two moving bodies (18 unknowns)"""
    w = _world(R, second="box_small.ztk")
    B, n = 4, 12
    rng = np.random.default_rng(3)
    dis = np.zeros((B, n)); vel = np.zeros((B, n))
    for b in range(B):
        dis[b, 2] = 0.05 - 1e-4
        dis[b, 6:9] = (rng.uniform(-0.02, 0.02), rng.uniform(-0.01, 0.01), 0.1 - 1e-4 + 0.025 - 1e-4); dis[b, 11] = rng.uniform(-0.5, 0.5)
        vel[b, 6] = rng.uniform(-0.3, 0.3)
    bt = R.Batch(w, B, max_rigid=4)
    os_ = []
    for b in range(B):
        o = oracle_cls(w.model); o.set_state(dis[b], vel[b]); o.update_init(); os_.append(o)
    errs = []; two = 0
    for k in range(150):
        sd = np.array([o.get_state()[0] for o in os_]); sv = np.array([o.get_state()[1] for o in os_])
        bt.set_state(sd, sv); bt.update_init(); bt.update(1)
        assert bt.status() == 0, R.last_error()
        d, v, a = bt.get_state()
        for b, o in enumerate(os_):
            assert o.update() == 0
            od, ov, oa = o.get_state()
            errs.append(max(np.abs(d[b] - od).max(), np.abs(v[b] - ov).max(), np.abs(a[b] - oa).max() / max(1.0, np.abs(oa).max())))
            two += len(o.volume_pairs()) >= 2
    errs = np.array(errs)
    assert two > 100
    assert np.median(errs) < 1e-9 and errs.max() < 1e-6, (np.median(errs), errs.max())


def test_volume_pair_capacity_overflow_is_reported(R):
    """three pairs in collision with room for two: status 2, not a silent drop"""
    w = _world(R, second="box_small.ztk")
    dis = np.zeros((1, 12)); dis[0, 2] = 0.05 - 1e-4; dis[0, 6:9] = (0.0, 0.0, 0.1 - 1e-4 + 0.025 - 1e-4)
    # the small box also reaches the floor when it hangs over the edge: put it beside the box, sunk into the floor
    dis2 = dis.copy(); dis2[0, 6:9] = (0.08, 0.0, 0.025 - 1e-4); dis2[0, 9:12] = 0
    bt = R.Batch(w, 1, max_rigid=1)
    bt.set_state(dis2, np.zeros((1, 12))); bt.update_init()
    assert bt.status() == 2


def test_volume_humanoid_standing_on_two_soles(R, oracle_cls):
    """the 30-DoF humanoid of config 4 under the Volume plugin: two rigid pairs (sole - floor), twelve unknowns, probe paths
    through the legs to the float base; 60 free-running steps of 6 instances against the oracle"""
    sc = R.scenarios.config4_volume(batch=6)
    bt = R.Batch(sc["world"], 6, max_rigid=sc["max_rigid"])
    bt.set_state(sc["dis"], sc["vel"]); bt.update_init(); bt.update(60)
    assert bt.status() == 0, R.last_error()
    d, v, a = bt.get_state()
    for b in range(6):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][b], sc["vel"][b]); o.update_init(); o.update_n(60)
        od, ov, oa = o.get_state()
        assert len(o.volume_pairs()) == 2
        assert np.abs(d[b] - od).max() < 1e-9 and np.abs(v[b] - ov).max() < 1e-7, (b, np.abs(d[b] - od).max(), np.abs(v[b] - ov).max())
        assert np.abs(a[b] - oa).max() / max(1.0, np.abs(oa).max()) < 1e-5


@pytest.mark.parametrize("root", ["fixed", "revolute"])
def test_volume_arm_presses_a_box(R, oracle_cls, root):
    """scenarios.arm_press under the Volume plugin: the hand of a 4-joint arm (DC motor with joint friction, torque motor,
    prismatic forearm) on a free box on the floor - probe paths that end at a fixed root / a 1-DoF root, a rigid pair
    between two moving bodies, driven motors.  40 free-running steps of 4 instances."""
    sc = R.scenarios.arm_press(batch=4, root=root, solver=R.SOLVER_VOLUME)
    bt = R.Batch(sc["world"], 4, max_rigid=6)
    bt.set_state(sc["dis"], sc["vel"]); bt.set_motor_input(sc["motor_in"]); bt.update_init(); bt.update(40)
    assert bt.status() == 0, R.last_error()
    d, v, a = bt.get_state()
    seen = 0
    for b in range(4):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][b], sc["vel"][b]); o.set_motor_input(sc["motor_in"][b]); o.update_init()
        for _ in range(40):
            assert o.update() == 0
            seen += len(o.volume_pairs()) >= 2
        od, ov, oa = o.get_state()
        assert np.abs(d[b] - od).max() < 1e-8 and np.abs(v[b] - ov).max() < 1e-6, (b, np.abs(d[b] - od).max(), np.abs(v[b] - ov).max())
    assert seen > 40


@pytest.mark.parametrize("cfg", ["config1_volume", "config4_volume"])
def test_volume_gpu_reproduces_golden(R, cfg):
    """the committed golden vectors of the Volume workloads (self-generated by the oracle, tests/golden/make_golden.py): states
    after 1 and 20 steps; no oracle needed at run time"""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = json.load(open(os.path.join(root, "tests", "golden", cfg + ".json")))
    B = len(g["instances"])
    sc = R.scenarios.CONFIGS[cfg](batch=B)
    b = R.Batch(sc["world"], B, device=0, max_rigid=sc["max_rigid"])
    b.set_state(np.asarray(g["dis0"]), np.asarray(g["vel0"]))
    b.update_init()
    assert b.status() == 0
    acc0 = b.get_state()[2]
    for i, rec in enumerate(g["instances"]):
        assert np.allclose(acc0[i], rec["acc_init"], rtol=1e-8, atol=1e-8)
    n = 0
    for cp in sorted(int(k) for k in g["instances"][0]["steps"]):
        b.update(cp - n); n = cp
        assert b.status() == 0
        d, v, a = b.get_state()
        for i, rec in enumerate(g["instances"]):
            exp = rec["steps"][str(cp)]
            assert len(exp["wrenches"]) > 0
            assert np.allclose(d[i], exp["dis"], rtol=1e-8, atol=1e-8)
            assert np.allclose(v[i], exp["vel"], rtol=1e-6, atol=1e-6)


def test_volume_rolling_cylinder(R, oracle_cls):
    """a 16-sided cylinder (models/cylinder.ztk: a curved primitive tessellated by the reader, 18 faces) pushed along the floor
    with half the spin of rolling: it slips (kinetic friction), then rolls from facet to facet (static friction, v = omega r
    within the polygon's bumps).  300 steps of 4 cylinders, every step started from the oracle's state.
    At the slip -> stick transition the centre of normal force sits on the edge of the contact polygon (the fix-up puts it there
    with a margin of zTOL = 1e-12) and the static-friction LP is decided by margins of that size: one step in a thousand
    comes out on the other side on the GPU - and equally in the oracle itself when its start state is perturbed by 1e-12
    (tools/vol_debug_state.py: 1 of 16 perturbed starts flips, on both sides).  Such steps are counted, not compared."""
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VOLUME)
    w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "cylinder.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    B = 4
    dis = np.zeros((B, 6)); vel = np.zeros((B, 6))
    for b in range(B):
        dis[b, 2] = 0.04 - 1e-5; dis[b, 5] = 0.3 * b; vel[b, 0] = 0.3 * np.cos(0.3 * b); vel[b, 1] = 0.3 * np.sin(0.3 * b)
        vel[b, 3] = -0.5 * 7.5 * np.sin(0.3 * b); vel[b, 4] = 0.5 * 7.5 * np.cos(0.3 * b)
    bt = R.Batch(w, B, max_rigid=1)
    os_ = []
    for b in range(B):
        o = oracle_cls(w.model); o.set_state(dis[b], vel[b]); o.update_init(); os_.append(o)
    errs = []; types = set()
    for k in range(300):
        sd = np.array([o.get_state()[0] for o in os_]); sv = np.array([o.get_state()[1] for o in os_])
        bt.set_state(sd, sv); bt.update_init(); bt.update(1)
        assert bt.status() == 0, R.last_error()
        d, v, a = bt.get_state()
        for b, o in enumerate(os_):
            assert o.update() == 0
            od, ov, oa = o.get_state()
            errs.append(max(np.abs(d[b] - od).max(), np.abs(v[b] - ov).max(), np.abs(a[b] - oa).max() / max(1.0, np.abs(oa).max())))
            for p in o.volume_pairs():
                if np.abs(p["wrench"]).max() > 0:
                    types.add(p["type"])
    errs = np.array(errs)
    assert types == {R.SF, R.KF}
    o = os_[0]; v = o.get_state()[1]
    assert abs(v[0] - v[4] * 0.04) < 0.02 * abs(v[0]) + 1e-3            # rolling
    flipped = int((errs > 1e-6).sum())
    assert flipped <= len(errs) // 100, flipped
    assert np.median(errs) < 1e-11 and np.sort(errs)[len(errs) - 1 - flipped] < 1e-6, (np.median(errs), flipped)


def test_volume_cylinder_on_its_end_sixteen_conditions(R, oracle_cls):
    """the 16-sided cylinder standing on its end: a contact polygon with 16 edges (17 QP constraints, a static-friction LP with
    128 + 6 columns: the three-columns-per-lane tableau); at rest, pushed (it tilts onto an edge of its end face) and twisted.
    100 steps, every step from the oracle's state; knife-edge steps are counted as in the rolling test."""
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VOLUME)
    w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "cylinder.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    B = 3
    dis = np.zeros((B, 6)); vel = np.zeros((B, 6))
    dis[:, 3] = np.pi / 2; dis[:, 2] = 0.06 - 1e-5
    vel[1, 0] = 0.2; vel[2, 5] = 3.0
    bt = R.Batch(w, B, max_rigid=1)
    os_ = []
    for b in range(B):
        o = oracle_cls(w.model); o.set_state(dis[b], vel[b]); o.update_init(); os_.append(o)
        assert len(o.volume_pairs()[0]["planes"]) == 16
    errs = []; n16 = 0
    for k in range(100):
        sd = np.array([o.get_state()[0] for o in os_]); sv = np.array([o.get_state()[1] for o in os_])
        bt.set_state(sd, sv); bt.update_init(); bt.update(1)
        assert bt.status() == 0, R.last_error()
        d, v, a = bt.get_state()
        for b, o in enumerate(os_):
            assert o.update() == 0
            od, ov, oa = o.get_state()
            errs.append(max(np.abs(d[b] - od).max(), np.abs(v[b] - ov).max(), np.abs(a[b] - oa).max() / max(1.0, np.abs(oa).max())))
            n16 += any(len(p["planes"]) >= 16 for p in o.volume_pairs())
    errs = np.array(errs)
    flipped = int((errs > 1e-6).sum())
    assert n16 > 100 and flipped <= 3, (n16, flipped)
    assert np.median(errs) < 1e-10 and np.sort(errs)[len(errs) - 1 - flipped] < 1e-6, (np.median(errs), flipped)


def test_volume_on_random_trees(R, oracle_cls, tmp_path):
    """random float-root trees (revolute / prismatic / fixed joints, random branching) carrying up to four boxes fall onto the
    floor under the Volume plugin: up to four rigid pairs at once on links of every depth - probe paths through random
    joint chains, several pairs sharing ancestors.  Chunks of 5 steps, each started from the oracle's state."""
    from randtree import random_tree_ztk
    rng = np.random.default_rng(78)
    npairs = 0; worst = 0.0; nbad = 0; ntot = 0
    for k in range(12):
        seed = 700 + k
        nlink = int(rng.integers(4, 16))
        f = tmp_path / f"rand{seed}.ztk"
        f.write_text(random_tree_ztk(seed, nlink, root="float", shapes=min(4, nlink)))
        w = R.World(solver=R.SOLVER_VOLUME)
        w.contact_info(os.path.join(R.scenarios.MODELS, "contactinfo.ztk"))
        h = w.reg_file(str(f)); w.reg_file(os.path.join(R.scenarios.MODELS, "floor.ztk"))
        m = w.model.contents
        B = 3
        dis = np.zeros((B, m.ndof)); vel = np.zeros((B, m.ndof))
        r2 = np.random.default_rng(seed)
        dis[:, 6:] = r2.uniform(-0.5, 0.5, (B, m.ndof - 6)); dis[:, 3:6] = r2.uniform(-0.3, 0.3, (B, 3))
        vel[:, 2] = -0.3; vel[:, 3:6] = r2.uniform(-1.0, 1.0, (B, 3))
        for i in range(B):
            dis[i, 2] -= R.scenarios.lowest_vertex_z(m, dis[i], h) - 0.002
        bt = R.Batch(w, B, max_rigid=6)
        orc = []
        for i in range(B):
            o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); orc.append(o)
        for chunk in range(16):
            sd = np.array([o.get_state()[0] for o in orc]); sv = np.array([o.get_state()[1] for o in orc])
            bt.set_state(sd, sv); bt.update_init(); bt.update(5)
            assert bt.status() == 0, (seed, R.last_error())
            d, v, a = bt.get_state()
            for i, o in enumerate(orc):
                for _ in range(5):
                    assert o.update() == 0
                    npairs += len(o.volume_pairs())
                od, ov, oa = o.get_state()
                e = max(np.abs(d[i] - od).max() / max(1.0, np.abs(od).max()), np.abs(v[i] - ov).max() / max(1.0, np.abs(ov).max()))
                worst = max(worst, e); ntot += 1; nbad += e > 1e-6
    assert npairs > 300
    # (a chunk with a knife-edge friction decision differs visibly, see test_volume_rolling_cylinder; the rest agrees closely)
    assert nbad <= ntot // 50, (nbad, ntot, worst)


@pytest.mark.parametrize("who", ["box", "floor", "both"])
def test_volume_slide_mode(R, oracle_cls, who):
    """cells in slide mode (fake crawler belts) under the Volume plugin: a box sliding over the floor, the box's bottom and / or
    the floor running; 120 steps of 3 instances, every step from the oracle's state"""
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VOLUME); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    bx = w.reg_file(os.path.join(M, "box.ztk")); fl = w.reg_file(os.path.join(M, "floor.ztk"))
    if who in ("box", "both"):
        w.set_slide(bx, 0, True, 0.2, (0.0, 1.0, 0.0))
    if who in ("floor", "both"):
        w.set_slide(fl, 0, True, -0.4, (0.3, 1.0, 0.0))
    B = 3
    dis = np.zeros((B, 6)); vel = np.zeros((B, 6))
    dis[:, 2] = 0.0499; dis[:, 5] = (0.2, -0.4, 1.0); vel[:, 0] = (0.3, 0.1, -0.2); vel[:, 1] = (0.0, 0.25, 0.1)
    bt = R.Batch(w, B, max_rigid=2)
    os_ = []
    for b in range(B):
        o = oracle_cls(w.model); o.set_state(dis[b], vel[b]); o.update_init(); os_.append(o)
    errs = []; kin = 0
    for k in range(120):
        sd = np.array([o.get_state()[0] for o in os_]); sv = np.array([o.get_state()[1] for o in os_])
        bt.set_state(sd, sv); bt.update_init(); bt.update(1)
        assert bt.status() == 0, R.last_error()
        d, v, a = bt.get_state()
        for b, o in enumerate(os_):
            assert o.update() == 0
            od, ov, oa = o.get_state()
            errs.append(max(np.abs(d[b] - od).max(), np.abs(v[b] - ov).max(), np.abs(a[b] - oa).max() / max(1.0, np.abs(oa).max())))
            kin += any(p["type"] == R.KF for p in o.volume_pairs())
    errs = np.array(errs)
    flipped = int((errs > 1e-6).sum())
    assert kin > 50 and flipped <= 3, (kin, flipped)
    assert np.median(errs) < 1e-10 and np.sort(errs)[len(errs) - 1 - flipped] < 1e-6, (np.median(errs), flipped)
