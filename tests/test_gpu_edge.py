"""GPU tier, edge cases of the boundary: fused multi-step launches, state round trips, motor
inputs, evaluation-only entry point, capacity overflow, small / degenerate worlds."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _rel(x, y):
    return np.abs(x - y).max() / max(1.0, np.abs(y).max())


def test_fused_steps_equal_single_steps(R):
    """rkfdBatchUpdate(n) == n x rkfdBatchUpdate(1), bit for bit (state round-trips through HBM)"""
    sc = R.scenarios.config4(batch=64)
    out = []
    for chunks in ([7], [1] * 7, [3, 4]):
        b = R.Batch(sc["world"], 64, max_rigid=sc["max_rigid"])
        b.set_state(sc["dis"], sc["vel"]); b.update_init()
        for n in chunks:
            b.update(n)
        assert b.status() == 0
        out.append(b.get_state() + b.get_contact())
    for other in out[1:]:
        for x, y in zip(out[0], other):
            assert np.array_equal(x, y)


def test_contact_and_pivot_state_round_trip(R, oracle_cls):
    """persistent state (stick anchors, stick/slip types, friction pivots) can be read out and
    put back: continuing from a restored batch equals continuing the original"""
    sc = R.scenarios.config4(batch=16)
    a = R.Batch(sc["world"], 16, max_rigid=sc["max_rigid"])
    a.set_state(sc["dis"], sc["vel"]); a.update_init(); a.update(10)
    dis, vel, _ = a.get_state(); act, typ, ref, _ = a.get_contact(); pt, pp = a.get_pivot()
    b = R.Batch(sc["world"], 16, max_rigid=sc["max_rigid"])
    b.set_state(dis, vel); b.set_contact(act, typ, ref); b.set_pivot(pt, pp)
    a.update(5); b.update(5)
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)


def test_motor_inputs_drive_the_joints(R, oracle_cls):
    """DC-motor voltages per instance (rkJointMotorSetInput): GPU vs oracle with saturating and
    non-saturating inputs, friction pivots flipping between stick and slip"""
    B = 8
    sc = R.scenarios.config3(batch=B)
    m = sc["world"].model.contents
    rng = np.random.default_rng(3)
    inp = np.zeros((B, m.nlink))
    jt = m.arr("jtype", m.nlink)
    inp[:, jt == R.JOINT_REVOL] = rng.uniform(-30, 30, (B, int((jt == R.JOINT_REVOL).sum())))   # beyond +-24 V too
    b = R.Batch(sc["world"], B, max_rigid=0)
    b.set_state(sc["dis"], sc["vel"]); b.set_motor_input(inp)
    b.update_init(); b.update(5)
    assert b.status() == 0
    dis, vel, acc = b.get_state(); pt, pp = b.get_pivot()
    for i in range(B):
        o = oracle_cls(sc["world"].model)
        o.set_state(sc["dis"][i], sc["vel"][i]); o.set_motor_input(inp[i]); o.update_init()
        for _ in range(5):
            o.update()
        od, ov, oa = o.get_state(); opt, opp = o.get_pivot()
        assert _rel(dis[i], od) < RTOL and _rel(vel[i], ov) < RTOL and _rel(acc[i], oa) < RTOL
        assert (pt[i][jt == R.JOINT_REVOL] == opt[jt == R.JOINT_REVOL]).all()
        assert _rel(pp[i][jt == R.JOINT_REVOL], opp[jt == R.JOINT_REVOL]) < RTOL
    assert np.abs(vel[:, 6:]).max() > 1e-3      # something actually moved


def test_eval_entry_point(R, oracle_cls):
    """rkfdBatchEval = one _rkFDUpdate / _rkFDUpdateRef at the current state: fills acc and the
    contact forces, leaves dis / vel untouched; only the committing variant moves friction pivots"""
    sc = R.scenarios.config4(batch=4)
    b = R.Batch(sc["world"], 4, max_rigid=sc["max_rigid"])
    b.set_state(sc["dis"], sc["vel"])
    b.eval(False)
    assert b.status() == 0
    dis, vel, acc = b.get_state()
    assert np.array_equal(dis, sc["dis"]) and np.array_equal(vel, sc["vel"])
    for i in range(4):
        o = oracle_cls(sc["world"].model)
        o.set_state(sc["dis"][i], sc["vel"][i]); o.eval(False)
        assert _rel(acc[i], o.get_state()[2]) < RTOL
        assert _rel(b.get_contact()[3][i], o.get_contact()[3]) < RTOL


def test_contact_capacity_overflow_is_reported(R):
    sc = R.scenarios.config1_rigid(batch=2)
    dis = sc["dis"].copy(); dis[:, 2] = 0.0499; dis[:, 3:] = 0      # box flat on the floor: 4 contacts
    b = R.Batch(sc["world"], 2, max_rigid=2)
    b.set_state(dis, sc["vel"]); b.update_init()
    assert b.status() == 2
    b2 = R.Batch(sc["world"], 2, max_rigid=4)
    b2.set_state(dis, sc["vel"]); b2.update_init()
    assert b2.status() == 0


def test_vert_plugin_without_rigid_capacity_reports_rigid_contact(R):
    """a batch created with no rigid-contact capacity (max_rigid = 0) has no rigid solver set up: a rigid
    contact is then an error status, never a silent wrong answer"""
    w = R.World(solver=R.SOLVER_VERT)
    w.contact_info(os.path.join(R.scenarios.MODELS, "contactinfo.ztk"))
    w.reg_file(os.path.join(R.scenarios.MODELS, "box.ztk")); w.reg_file(os.path.join(R.scenarios.MODELS, "floor.ztk"))
    b = R.Batch(w, 1, max_rigid=0)
    b.set_state(np.array([[0, 0, 0.0499, 0, 0, 0.0]]), np.zeros((1, 6))); b.update_init()
    assert b.status() != 0


def test_batch_of_one_and_world_without_contacts(R, oracle_cls):
    sc = R.scenarios.config2(batch=1)
    b = R.Batch(sc["world"], 1, max_rigid=0)
    b.set_state(sc["dis"], sc["vel"]); b.update_init(); b.update(3)
    o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][0], sc["vel"][0]); o.update_init()
    for _ in range(3):
        o.update()
    assert _rel(b.get_state()[0][0], o.get_state()[0]) < RTOL
    act, typ, ref, f = b.get_contact()
    assert act.shape == (1, 0)


def test_limits_are_reported(R):
    """a world beyond the per-wave limits fails at creation with a message"""
    w = R.World(solver=R.SOLVER_VERT)
    for _ in range(3):
        w.reg_file(os.path.join(R.scenarios.MODELS, "chain30.ztk"))     # 90 joint coordinates > 64
    with pytest.raises(R.RkfdError, match="exceeds"):
        R.Batch(w, 1, max_rigid=0)
    sc = R.scenarios.config4(batch=1)
    with pytest.raises(R.RkfdError, match="exceeds"):
        R.Batch(sc["world"], 1, max_rigid=50)


@pytest.mark.parametrize("root,with_box", [("fixed", True), ("revolute", True), ("fixed", False)])
def test_arm_press_paths(R, oracle_cls, root, with_box):
    """MLCP branches the humanoid workloads do not take: contact paths ending at a fixed / 1-DoF root, a
    rigid pair with two moving sides (hand on a free box), prismatic joint, both motor types driven"""
    B, nsteps = 8, 30
    sc = R.scenarios.arm_press(batch=B, root=root, with_box=with_box)
    b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    b.set_state(sc["dis"], sc["vel"]); b.set_motor_input(sc["motor_in"])
    b.update_init(); b.update(nsteps)
    assert b.status() == 0
    dis, vel, acc = b.get_state()
    act, typ, ref, f = b.get_contact()
    seen = 0
    for i in range(B):
        o = oracle_cls(sc["world"].model)
        o.set_state(sc["dis"][i], sc["vel"][i]); o.set_motor_input(sc["motor_in"][i]); o.update_init()
        o.update_n(nsteps)
        od, ov, oa = o.get_state()
        assert _rel(dis[i], od) < RTOL and _rel(vel[i], ov) < RTOL and _rel(acc[i], oa) < 1e-8
        oact, otyp, oref, of = o.get_contact()
        assert (act[i] == oact).all()
        assert _rel(f[i], of * (oact[:, None] != 0)) < 1e-8
        seen += int(oact.sum())
    if with_box:
        assert seen > 0


@pytest.mark.parametrize("solver", ["mlcp", "vert"])
def test_self_collision(R, oracle_cls, solver):
    """SELF-COLLISION on the GPU (scenarios.arm_fold: the folded arm's last link presses on its first): a rigid contact
    whose two sides are links of ONE chain - the pairs registration forms by default (reference src/rkfd_sim.c:198; probed
    through the "self collision" branch of src/rkfd_util.c:163-170).  The finger bounces and the system is stiff by
    construction (the oracle's own d(acc)/d(q) is 5e5 ... 8e7 s^-2 in contact), so every step starts from the oracle's
    state and the tolerance is 1e-9 relative plus what a 2e-14 rad nudge of the start state does to the oracle's own
    result (same test under the lane emulator: tests/test_emu_parity.py)."""
    B, nsteps = 8, 12
    sc = R.scenarios.arm_fold(batch=B, solver=R.SOLVER_MLCP if solver == "mlcp" else R.SOLVER_VERT)
    model = sc["world"].model
    b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    b.set_motor_input(sc["motor_in"])
    b.set_state(sc["dis"], sc["vel"]); b.update_init()
    os_ = []
    for i in range(B):
        o = oracle_cls(model)
        o.set_state(sc["dis"][i], sc["vel"][i]); o.set_motor_input(sc["motor_in"][i]); o.update_init()
        os_.append(o)
    in_contact = 0

    def step_from(i, st, ct, pv, dq):
        o = oracle_cls(model); o.set_motor_input(sc["motor_in"][i])
        d = st[0].copy(); d[dq[0]] += dq[1]
        o.set_state(d, st[1]); o.set_contact(*ct[:3]); o.set_pivot(*pv); o.update_init()
        o.set_contact(*ct[:3]); o.set_pivot(*pv)
        o.update()
        return o
    for k in range(nsteps):
        st = [o.get_state() for o in os_]; ct = [o.get_contact() for o in os_]; pv = [o.get_pivot() for o in os_]
        b.set_state(np.array([x[0] for x in st]), np.array([x[1] for x in st]))
        b.set_contact(np.array([c[0] for c in ct]), np.array([c[1] for c in ct]), np.array([c[2] for c in ct]))
        b.set_pivot(np.array([p_[0] for p_ in pv]), np.array([p_[1] for p_ in pv]))
        b.update(1)
        assert b.status() == 0
        dis, vel, acc = b.get_state()
        act, typ, ref, f = b.get_contact()
        for i, o in enumerate(os_):
            o.update()
            od, ov, oa = o.get_state()
            oact, otyp, oref, of = o.get_contact()
            in_contact += int(oact.sum() >= 1 and np.abs(of).max() > 1.0)
            sa = sf = 0.0
            for j in range(3):
                on = step_from(i, st[i], ct[i], pv[i], (j, 1e-12))
                sa = max(sa, np.abs(on.get_state()[2] - oa).max() / 1e-12); sf = max(sf, np.abs(on.get_contact()[3] - of).max() / 1e-12)
            assert np.abs(dis[i] - od).max() < 1e-9 and np.abs(vel[i] - ov).max() < 1e-9 * max(1.0, np.abs(ov).max()) + 2e-17 * sa
            assert np.abs(acc[i] - oa).max() < 1e-9 * max(1.0, np.abs(oa).max()) + 2e-14 * sa, (k, i, sa)
            assert (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all()
            assert np.abs(f[i] - of * (oact[:, None] != 0)).max() < 1e-9 * max(1.0, np.abs(of).max()) + 2e-14 * sf, (k, i, sf)
    assert in_contact >= B * nsteps // 2
    # the arm's own pairs unregistered, as the reference's arm drivers do (arm_box_test.c:49): the finger meets nothing
    sc2 = R.scenarios.arm_fold(batch=2, unreg=True)
    b2 = R.Batch(sc2["world"], 2, max_rigid=sc2["max_rigid"])
    b2.set_state(sc2["dis"], sc2["vel"]); b2.set_motor_input(sc2["motor_in"]); b2.update_init(); b2.update(3)
    assert b2.status() == 0 and b2.get_contact()[0].sum() == 0
    for i in range(2):
        o = oracle_cls(sc2["world"].model)
        o.set_state(sc2["dis"][i], sc2["vel"][i]); o.set_motor_input(sc2["motor_in"][i]); o.update_init(); o.update_n(3)
        assert _rel(b2.get_state()[2][i], o.get_state()[2]) < RTOL


def test_far_from_the_world_origin(R, oracle_cls):
    """spatial quantities are taken about the anchor link, not the world origin: a humanoid in free
    flight 1 km away from the origin still matches the oracle (parallel-axis terms about the world
    origin would cost ~(distance / link size)^2 in relative accuracy)"""
    sc = R.scenarios.config4(batch=4)
    dis = sc["dis"].copy(); dis[:, 0] += 1000.0; dis[:, 1] -= 500.0; dis[:, 2] += 0.5
    vel = np.random.default_rng(0).uniform(-1, 1, sc["vel"].shape)
    b = R.Batch(sc["world"], 4, max_rigid=sc["max_rigid"])
    b.set_state(dis, vel); b.update_init(); b.update(5)
    assert b.status() == 0
    d, v, a = b.get_state()
    for i in range(4):
        o = oracle_cls(sc["world"].model); o.set_state(dis[i], vel[i]); o.update_init(); o.update_n(5)
        od, ov, oa = o.get_state()
        assert _rel(a[i], oa) < 1e-11 and _rel(v[i], ov) < 1e-12
        assert np.abs(d[i] - od).max() < 1e-9          # absolute: |dis| ~ 1e3


def _stack_world(R):
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_MLCP); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    for f in ("box.ztk", "box_small.ztk", "box_small.ztk", "floor.ztk"):
        w.reg_file(os.path.join(M, f))
    m = w.model.contents
    dis = np.zeros((1, m.ndof)); vel = np.zeros((1, m.ndof))
    dis[0, 0:3] = (0, 0, 0.05 - 0.0005)                                  # box on the floor
    dis[0, 6:9] = (0.01, 0.0, 0.1 + 0.025 - 0.001); dis[0, 9:12] = (0, 0, 0.3)       # small box on the box, yawed
    dis[0, 12:15] = (-0.02, 0.01, 0.15 + 0.025 - 0.0015); dis[0, 15:18] = (0.1, 0, 0)   # small box on the small box
    vel[0, 6] = 0.2
    return w, dis, vel


def test_stacked_boxes(R, oracle_cls):
    """three free boxes stacked on the floor: rigid pairs with two moving sides whose paths end at
    different float joints, 96 candidate vertices (two collision chunks: anchors re-slotted through
    the copy), 30 steps of settling vs the oracle"""
    w, dis, vel = _stack_world(R)
    B = 6
    dis = np.repeat(dis, B, 0); vel = np.repeat(vel, B, 0)
    vel[:, 6] = np.linspace(0.0, 0.5, B)
    b = R.Batch(w, B, max_rigid=24)
    b.set_state(dis, vel); b.update_init()
    orc = []
    for i in range(B):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); orc.append(o)
    seen = 0
    for chunk in range(3):
        b.update(10)
        assert b.status() == 0
        d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
        for i, o in enumerate(orc):
            o.update_n(10)
            od, ov, oa = o.get_state(); oact, _, _, of = o.get_contact()
            assert (act[i] == oact).all()
            assert _rel(d[i], od) < 1e-9 and _rel(v[i], ov) < 1e-9 and _rel(a[i], oa) < 1e-8
            assert _rel(f[i], of * (oact[:, None] != 0)) < 1e-8
            seen += int(oact.sum())
    assert seen > 10


def test_elastic_and_rigid_contacts_in_one_evaluation(R, oracle_cls):
    """a box lying across the hard / soft seam of floor_hardsoft under the MLCP plugin: penalty wrenches
    (elastic vertices) enter the free accelerations, the rigid vertices are solved on top of them"""
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_MLCP); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "box.ztk")); w.reg_file(os.path.join(M, "floor_hardsoft.ztk"))
    m = w.model.contents
    B = 4
    dis = np.zeros((B, m.ndof)); vel = np.zeros((B, m.ndof))
    dis[:, 2] = 0.05 - 0.0005; dis[:, 5] = np.linspace(0.0, 0.6, B)
    vel[:, 0] = 0.3; vel[:, 5] = 1.0
    b = R.Batch(w, B, max_rigid=8)
    b.set_state(dis, vel); b.update_init(); b.update(40)
    assert b.status() == 0
    d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
    cit = m.arr("ci_type", m.nci); cci = m.arr("pair_ci", m.npair)
    for i in range(B):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); o.update_n(40)
        od, ov, oa = o.get_state(); oact, _, _, of = o.get_contact()
        assert (act[i] == oact).all()
        assert _rel(d[i], od) < RTOL and _rel(v[i], ov) < RTOL and _rel(a[i], oa) < RTOL
        assert _rel(f[i], of * (oact[:, None] != 0)) < RTOL
    assert {int(t) for t in cit[cci]} == {R.CONTACT_RIGID, R.CONTACT_ELASTIC}       # both kinds of pairs are registered


def _vert_box_world(R):
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VERT); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "box.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    return w


def test_vert_plugin_rigid_qp_box(R, oracle_cls):
    """the reference's DEFAULT plugin with a rigid pair: friction pyramids + active-set QP
    (reference src/rkfd_vert.c:258-336, src/rkfd_opt_qp.c).  A box dropped flat / tilted / sliding onto
    the rigid floor: apex-degenerate bases (all faces of a pyramid active), slipping and sticking
    vertices, make and break; 40 steps vs the oracle, whose KKT solves use a generic pseudo-inverse"""
    w = _vert_box_world(R)
    B = 8
    dis = np.zeros((B, 6)); vel = np.zeros((B, 6))
    dis[:, 2] = 0.0499
    dis[1:, 3:6] = np.random.default_rng(1).uniform(-0.3, 0.3, (B - 1, 3))
    vel[:, 0] = np.linspace(0.0, 0.4, B); vel[2:, 3:6] = np.random.default_rng(2).uniform(-1, 1, (B - 2, 3))
    m = w.model.contents
    for i in range(1, B):
        dis[i, 2] -= R.scenarios.lowest_vertex_z(m, dis[i], 0) + 0.0001
    b = R.Batch(w, B, max_rigid=8)
    b.set_state(dis, vel); b.update_init()
    orc = []
    for i in range(B):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); orc.append(o)
    seen = kf = 0
    for chunk in range(4):
        b.update(10)
        assert b.status() == 0
        d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
        for i, o in enumerate(orc):
            o.update_n(10)
            od, ov, oa = o.get_state(); oact, otyp, _, of = o.get_contact()
            assert (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all(), (i, chunk)
            assert _rel(d[i], od) < 1e-8 and _rel(v[i], ov) < 1e-8 and _rel(a[i], oa) < 1e-7, (i, chunk)
            assert _rel(f[i], of * (oact[:, None] != 0)) < 1e-7, (i, chunk)
            seen += int(oact.sum()); kf += int((otyp[oact != 0] == R.KF).sum())
    assert seen > 20 and kf > 0


def _fma_oracle_lib():
    """the rounding control: the oracle source built with fused multiply-adds (make -C oracle fma)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle"), "fma"], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(root, "oracle", "_build", "librkfd_oracle_fma.so")


def test_vert_plugin_rigid_qp_humanoid(R, oracle_cls):
    """config 4 under the default plugin: the humanoid standing on 8 sole vertices = 24 unknowns and 64 pyramid faces
    (one per lane) in the QP, then rocking on 3-7 (60 steps) vs the oracle, whose KKT systems go up to 88 x 88 through
    the generic pseudo-inverse.  The device is re-synchronised to the oracle after every step, so every one of the
    B x 60 instance-steps is a one-step comparison from identical inputs.
    Agreement is required to 1e-8 with identical contact sets / types on all but a few instance-steps: the active-set
    method (reference src/rkfd_opt_qp.c:43-181) decides with absolute 1e-12 tests on a QP whose Hessian A'A + L has a
    condition number above 1e10 here, and at exact ties two correct implementations take different branches.  The
    CONTROL in this test measures that rate for the algorithm itself: the same oracle source built with fused
    multiply-adds, re-synchronised the same way, disagrees with the plain build on 4 of 480 instance-steps (measured;
    the HIP path on 4 of 480, three of them the very same steps)."""
    B, N = 8, 60
    sc = R.scenarios.config4_vert(batch=B)
    fma = _fma_oracle_lib()
    b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    b.set_state(sc["dis"], sc["vel"]); b.update_init()
    orc, ctl = [], []
    for i in range(B):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init(); orc.append(o)
        c = oracle_cls(sc["world"].model, fma); c.set_state(sc["dis"][i], sc["vel"][i]); c.update_init(); ctl.append(c)
    seen = dev_bad = ctl_bad = 0
    m = sc["world"].model.contents
    for s in range(N):
        b.update(1)
        assert b.status() == 0
        d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
        od = np.zeros_like(d); ov = np.zeros_like(v)
        oact = np.zeros_like(act); otyp = np.zeros_like(typ); oref = np.zeros_like(ref)
        optyp = np.zeros((B, m.nlink), dtype=np.int32); opprev = np.zeros((B, m.nlink))
        for i, (o, c) in enumerate(zip(orc, ctl)):
            assert o.update() == 0 and c.update() == 0
            od[i], ov[i], oa = o.get_state(); oact[i], otyp[i], oref[i], of = o.get_contact()
            optyp[i], opprev[i] = o.get_pivot()
            ok = (act[i] == oact[i]).all() and (typ[i] == otyp[i] * (oact[i] != 0)).all() and \
                max(_rel(d[i], od[i]), _rel(v[i], ov[i]), _rel(f[i], of * (oact[i][:, None] != 0))) < 1e-8 and _rel(a[i], oa) < 1e-7
            dev_bad += 0 if ok else 1
            cd, cv, ca = c.get_state(); cact, ctyp, cref, cf = c.get_contact()
            okc = (cact == oact[i]).all() and (ctyp == otyp[i]).all() and max(_rel(cv, ov[i]), _rel(cf, of)) < 1e-8
            ctl_bad += 0 if okc else 1
            c.set_state(od[i], ov[i]); c.set_contact(oact[i], otyp[i], oref[i]); c.set_pivot(optyp[i], opprev[i])
            seen += int(oact[i].sum())
        b.set_state(od, ov); b.set_contact(oact, otyp, oref); b.set_pivot(optyp, opprev)
    assert seen > 5 * B * N                          # sustained multi-vertex contact throughout
    assert dev_bad <= max(2 * ctl_bad, 6) and dev_bad <= 0.03 * B * N, (dev_bad, ctl_bad)


@pytest.mark.parametrize("P,cap", [(4, 8), (6, 8), (12, 5), (16, 4)])
def test_vert_plugin_other_pyramids(R, oracle_cls, P, cap):
    """rkFDPrpSetPyramid: pyramids with 4 / 6 / 12 / 16 faces (here within 64 faces: one face per lane), a tilted
    sliding box on the rigid floor vs the oracle; the capacity ends at 192 faces (three 64-bit words of active flags)"""
    w = _vert_box_world(R); w.set_pyramid(P)
    B = 4
    dis = np.zeros((B, 6)); vel = np.zeros((B, 6))
    dis[:, 2] = 0.0499; dis[1:, 3:6] = np.random.default_rng(3).uniform(-0.25, 0.25, (B - 1, 3))
    vel[:, 0] = np.linspace(0.1, 0.5, B); vel[:, 1] = 0.07
    m = w.model.contents
    for i in range(1, B):
        dis[i, 2] -= R.scenarios.lowest_vertex_z(m, dis[i], 0) + 0.0001
    b = R.Batch(w, B, max_rigid=cap)
    b.set_state(dis, vel); b.update_init(); b.update(30)
    assert b.status() == 0
    d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
    for i in range(B):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); o.update_n(30)
        od, ov, oa = o.get_state(); oact, otyp, _, of = o.get_contact()
        assert (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all()
        assert _rel(d[i], od) < 1e-8 and _rel(v[i], ov) < 1e-8 and _rel(a[i], oa) < 1e-6
    with pytest.raises(R.RkfdError, match="pyramid|per-wave limit"):
        R.Batch(w, 1, max_rigid=192 // P + 1)


def test_vert_plugin_wide_qp_box(R, oracle_cls):
    """more pyramid faces than lanes: capacity 16 vertices x 8 faces = 128 (rkfd_vert_qp_wide, vert_rigid == 3; the reference has
    no such limit, src/rkfd_vert.c:73-103).  Tilted / sliding / spinning boxes vs the oracle, and against the same boxes run
    through the one-face-per-lane form (capacity 8): same contact sets and friction states, values to rounding."""
    w = _vert_box_world(R)
    B = 16
    rng = np.random.default_rng(5)
    dis = np.zeros((B, 6)); vel = np.zeros((B, 6))
    dis[:, 2] = 0.0499
    dis[1:, 3:6] = rng.uniform(-0.3, 0.3, (B - 1, 3))
    vel[:, 0] = np.linspace(0.0, 0.4, B); vel[2:, 3:6] = rng.uniform(-1, 1, (B - 2, 3))
    m = w.model.contents
    for i in range(1, B):
        dis[i, 2] -= R.scenarios.lowest_vertex_z(m, dis[i], 0) + 0.0001
    out = []
    for cap in (16, 8):
        b = R.Batch(w, B, max_rigid=cap)
        b.set_state(dis, vel); b.update_init(); b.update(30)
        assert b.status() == 0
        out.append(b.get_state() + b.get_contact())
    d, v, a, act, typ, ref, f = out[0]
    agree = 0
    for i in range(B):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); o.update_n(30)
        od, ov, oa = o.get_state(); oact, otyp, _, of = o.get_contact()
        # (the method's 1e-12 knife-edge decisions: an instance may leave the oracle's active-set path - tools/vert_agreement.py;
        #  the one-face-per-lane form shows the same rate)
        if (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all() and _rel(d[i], od) < 1e-8 and _rel(v[i], ov) < 1e-8:
            agree += 1
    assert agree >= B - 2, agree
    same = sum(int((out[0][3][i] == out[1][3][i]).all() and _rel(out[0][0][i], out[1][0][i]) < 1e-8) for i in range(B))
    assert same >= B - 2, same


def test_vert_plugin_config5_world(R, oracle_cls):
    """config 5 under the reference's DEFAULT plugin (VERDICT r02 missing 4): humanoid + four boxes on 24 contact vertices =
    72 unknowns and 192 pyramid faces, the wide form of the QP; 10 steps vs the oracle"""
    B, N = 8, 10
    sc = R.scenarios.config5_vert(batch=B)
    b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    assert b.lds_bytes > 64 * 1024                     # (one instance per CU: the QP's matrices)
    b.set_state(sc["dis"], sc["vel"]); b.update_init(); b.update(N)
    assert b.status() == 0
    d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
    assert act.sum(1).min() >= 20
    for i in range(0, B, 3):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init(); o.update_n(N)
        od, ov, oa = o.get_state(); oact, otyp, _, of = o.get_contact()
        assert (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all()
        assert _rel(d[i], od) < 1e-9 and _rel(v[i], ov) < 1e-8 and _rel(a[i], oa) < 1e-6
        assert _rel(f[i], of * (oact[:, None] != 0)) < 1e-5      # (24 coplanar vertices: the force split is ill-conditioned, the wrench is not)


@pytest.mark.parametrize("plugin,need", [("mlcp", 64), ("vert", 52)])
def test_agreement_rate_on_random_box_drops(R, oracle_cls, plugin, need):
    """64 random box drops (tilted / flat, sliding, spinning) x 40 steps: how many instances stay on the
    oracle's path (contact sets, stick/slip types, velocities to 1e-6).  MLCP: all.  Vert: the active-set
    method decides with absolute 1e-12 tests on an ill-conditioned QP, so two correct implementations can
    branch differently at exact ties; measured 58-60 of 64, required > 80 %.
    CONTROL: the same oracle source built with fused multiply-adds runs the same drops; how many of ITS instances
    stay on the plain build's path is the rate the algorithm itself allows, and the HIP path may not do worse than
    that by more than a few instances."""
    M = R.scenarios.MODELS
    N, S = 64, 40
    w = R.World(solver=R.SOLVER_VERT if plugin == "vert" else R.SOLVER_MLCP); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "box.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    m = w.model.contents
    rng = np.random.default_rng(11)
    dis = np.zeros((N, 6)); vel = np.zeros((N, 6))
    dis[:, 3:6] = rng.uniform(-0.4, 0.4, (N, 3)) * (rng.random((N, 1)) < 0.8)
    vel[:, 0:2] = rng.uniform(-0.5, 0.5, (N, 2)); vel[:, 3:6] = rng.uniform(-1.5, 1.5, (N, 3)) * (rng.random((N, 1)) < 0.5)
    for i in range(N):
        dis[i, 2] = 0.2
        dis[i, 2] = 0.2 - R.scenarios.lowest_vertex_z(m, dis[i], 0) - 0.0002
    b = R.Batch(w, N, max_rigid=8); b.set_state(dis, vel); b.update_init()
    fma = _fma_oracle_lib()
    orc, ctl = [], []
    for i in range(N):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); orc.append(o)
        c = oracle_cls(w.model, fma); c.set_state(dis[i], vel[i]); c.update_init(); ctl.append(c)
    alive = np.ones(N, dtype=bool); calive = np.ones(N, dtype=bool)
    for s in range(S):
        b.update(1); d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
        for i, (o, c) in enumerate(zip(orc, ctl)):
            o.update(); c.update()
            od, ov, oa = o.get_state(); oact, otyp, _, _ = o.get_contact()
            if alive[i]:
                alive[i] = bool((act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all()
                                and np.abs(v[i] - ov).max() < 1e-6 * max(1.0, np.abs(ov).max()))
            if calive[i]:
                cd, cv, ca = c.get_state(); cact, ctyp, _, _ = c.get_contact()
                calive[i] = bool((cact == oact).all() and (ctyp == otyp).all()
                                 and np.abs(cv - ov).max() < 1e-6 * max(1.0, np.abs(ov).max()))
    assert b.status() == 0
    assert alive.sum() >= need, int(alive.sum())
    assert alive.sum() >= calive.sum() - 4, (int(alive.sum()), int(calive.sum()))


@pytest.mark.parametrize("solver,floor,who", [("mlcp", "floor.ztk", "box"), ("vert", "floor.ztk", "floor"), ("vert", "floor_hardsoft.ztk", "box"), ("mlcp", "floor.ztk", "both")])
def test_slide_mode(R, oracle_cls, solver, floor, who):
    """cells in slide mode (fake crawler): belt on the box, on the floor, on both; MLCP, Vert QP and penalty
    contacts; 60 steps vs the oracle incl. the drifting stick anchors"""
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_MLCP if solver == "mlcp" else R.SOLVER_VERT); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    bx = w.reg_file(os.path.join(M, "box.ztk")); fl = w.reg_file(os.path.join(M, floor))
    if who in ("box", "both"):
        w.set_slide(bx, 0, True, 0.2, (0.0, 1.0, 0.0))
    if who in ("floor", "both"):
        w.set_slide(fl, 0, True, -0.1, (0.3, 1.0, 0.0))
    B = 4
    dis = np.zeros((B, 6)); vel = np.zeros((B, 6)); dis[:, 2] = 0.0499; dis[:, 5] = np.linspace(0.0, 0.6, B)
    if floor == "floor_hardsoft.ztk":
        dis[:, 1] = -1.0
    b = R.Batch(w, B, max_rigid=8); b.set_state(dis, vel); b.update_init(); b.update(60)
    assert b.status() == 0
    d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
    for i in range(B):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); o.update_n(60)
        od, ov, oa = o.get_state(); oact, otyp, oref, of = o.get_contact(); on = oact != 0
        assert (act[i] == oact).all() and (typ[i] == otyp * on).all()
        assert _rel(d[i], od) < 1e-8 and _rel(v[i], ov) < 1e-8
        assert np.abs(ref[i] - oref * on[:, None]).max() < 1e-8


def test_split_launches_give_the_same_results(R):
    """rkfdBatchSetSplit: the batch launched as 1, 2, 3, 4, 8 kernels on internal streams - bit-identical states
    (instances are independent), with and without intermediate joins; accessors and status wait for the parts"""
    sc = R.scenarios.config4(batch=300)          # not a multiple of the split counts
    out = []
    for K in (1, 2, 3, 4, 8):
        b = R.Batch(sc["world"], 300, max_rigid=sc["max_rigid"]); b.set_split(K)
        b.set_state(sc["dis"], sc["vel"]); b.update_init()
        for s in range(12):
            b.update(1 if s % 3 else 2)
            if s == 5:
                b.join(); assert b.status() == 0
        out.append(b.get_state() + b.get_contact() + b.get_pivot())
        assert b.status() == 0
    for o in out[1:]:
        for x, y in zip(out[0], o):
            assert np.array_equal(x, y)


@pytest.mark.gpu
def test_residency_report_follows_the_lds_allocation_pieces(R):
    """rkfdBatchResidency: the HIP occupancy answer corrected for the 1280-byte pieces in which the hardware hands out
    LDS (128 per CU; measured with tools/ubench/residency.hip) - and the benchmark worlds sit where DESIGN.md says"""
    for name, expect in (("config2", 11), ("config3", 10), ("config4", 11), ("config4v", 8), ("config5", 3)):
        sc = R.scenarios.CONFIGS[name](batch=8)
        b = R.Batch(sc["world"], 8, max_rigid=sc["max_rigid"])
        pieces = -(-b.lds_bytes // 1280)
        assert b.residency() == min(12, 128 // pieces) == expect, (name, b.lds_bytes, b.residency())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["config2", "config3", "config4", "config4v", "config5"])
def test_specialized_kernel_gives_the_same_results(R, cfg):
    """rkfdBatchSpecialize: the step kernel compiled for the world (hipRTC, its dimensions as literals) against the
    generic kernel - states, contact state and pivots bit for bit, every kernel variant (PGS full / packed, Vert QP)"""
    sc = R.scenarios.CONFIGS[cfg](batch=64)
    out = []
    for spec in (False, True):
        b = R.Batch(sc["world"], 64, max_rigid=sc["max_rigid"])
        if spec:
            b.specialize()
        b.set_state(sc["dis"], sc["vel"]); b.update_init()
        b.update(20); b.set_split(3); b.update(10)
        assert b.status() == 0
        out.append(b.get_state() + b.get_contact() + b.get_pivot())
    for x, y in zip(out[0], out[1]):
        assert np.array_equal(x, y)


def test_more_than_256_candidates(R, oracle_cls):
    """worlds whose candidate sweep takes more than four chunks of 64: the tessellated ball (274 candidates) sliding, sticking
    and rolling from vertex to vertex for 150 steps, and the humanoid with six sphere shells (764 candidates) standing for
    10 - against the oracle"""
    sc = R.scenarios.ball_roll(batch=8)
    b = R.Batch(sc["world"], 8, max_rigid=sc["max_rigid"]); b.set_state(sc["dis"], sc["vel"]); b.update_init()
    orc = []
    for i in range(8):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init(); orc.append(o)
    changes = 0; prev = b.get_contact()[0]
    for s in range(150):
        b.update(1)
        assert b.status() == 0
        d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
        for i, o in enumerate(orc):
            o.update(); od, ov, oa = o.get_state(); oact, otyp, _, of = o.get_contact()
            assert (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all(), (s, i)
            assert _rel(d[i], od) < 1e-7 and _rel(v[i], ov) < 1e-7 and _rel(f[i], of) < 1e-6, (s, i)
        changes += int((act != prev).sum()); prev = act
    assert changes > 8                                    # the contact moved from vertex to vertex
    sc = R.scenarios.config4_shell(batch=8)
    assert sc["world"].model.contents.ncand == 764
    b = R.Batch(sc["world"], 8, max_rigid=sc["max_rigid"]); b.set_state(sc["dis"], sc["vel"]); b.update_init(); b.update(10)
    assert b.status() == 0
    d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
    for i in range(8):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init(); o.update_n(10)
        od, ov, oa = o.get_state(); oact, otyp, _, of = o.get_contact()
        assert (act[i] == oact).all() and oact.sum() >= 5
        assert _rel(d[i], od) < 1e-8 and _rel(v[i], ov) < 1e-8 and _rel(a[i], oa) < 1e-8 and _rel(f[i], of) < 1e-8


@pytest.mark.parametrize("contact", [False, True], ids=["free", "contact"])
def test_spherical_joints(R, oracle_cls, contact):
    """models/arm_spher.ztk: three links on spherical joints (device: three pseudo-links per joint; oracle: a genuine 3-DoF
    joint) swinging freely for 200 steps, and pressing the hand on the rigid floor (contact paths through the
    pseudo-links) - against the oracle"""
    B = 16
    sc = R.scenarios.arm_spher(batch=B, contact=contact)
    b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"]); b.set_state(sc["dis"], sc["vel"]); b.update_init()
    orc = []
    for i in range(B):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init(); orc.append(o)
    seen = 0
    for s in range(8 if contact else 4):
        n = 1 if contact else 50
        b.update(n)
        assert b.status() == 0
        d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
        for i, o in enumerate(orc):
            o.update_n(n); od, ov, oa = o.get_state(); oact, otyp, _, of = o.get_contact()
            assert (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all()
            tol = 1e-7 if contact else 1e-9
            assert _rel(d[i], od) < tol and _rel(v[i], ov) < tol and _rel(a[i], oa) < tol and _rel(f[i], of) < tol, (s, i)
            seen += int(oact.sum())
    assert (seen > 0) == contact


def test_node_level_matches_one_batch(R):
    """the node level of the C ABI (rkfdNode*: every GPU of a node from one process, one host thread and stream per device, no
    per-step communication, one RCCL all-gather of the final states): 37 instances of config 4 as THREE uneven shards
    (13 / 12 / 12 - on one GPU box the three shards share device 0: the sharding, the threads, the packing and the plain
    copies are the ones an 8-GPU node runs) give bit for bit the states of one batch of 37; the all-gather itself runs through
    librccl on a one-device node (RCCL refuses two ranks on one device)."""
    B, nsteps = 37, 6
    sc = R.scenarios.config4(batch=B)
    b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    b.set_state(sc["dis"], sc["vel"]); b.update_init(); b.update(nsteps)
    assert b.status() == 0
    d0, v0, a0 = b.get_state()
    n3 = R.Node(sc["world"], B, max_rigid=sc["max_rigid"], devices=[0, 0, 0])
    assert [(lo, hi) for _d, lo, hi in n3.shards()] == [(0, 13), (13, 25), (25, 37)]
    n3.set_state(sc["dis"], sc["vel"]); n3.update_init(); n3.update(nsteps)
    assert n3.status() == 0
    d3, v3, a3 = n3.get_state()
    assert (d3 == d0).all() and (v3 == v0).all() and (a3 == a0).all()
    with pytest.raises(R.RkfdError, match="ncclCommInitAll"):
        n3.gather()                                    # three ranks on one device: refused by RCCL, reported, nothing hangs
    n3.close()
    n1 = R.Node(sc["world"], B, max_rigid=sc["max_rigid"], ndev=1)
    n1.specialize(); n1.set_split(3)
    n1.set_state(sc["dis"], sc["vel"]); n1.update_init(); n1.snapshot(); n1.update(nsteps)
    gd, gv = n1.gather()                               # ncclAllGather (one rank), then device 0's copy to the host
    assert n1.status() == 0
    assert (gd == d0).all() and (gv == v0).all()
    n1.restore(); n1.update(nsteps)                    # a second rollout from the snapshot ends in the same states
    gd2, gv2 = n1.gather()
    assert (gd2 == d0).all() and (gv2 == v0).all()
    # launch shaping from the node level: lane mapping by measurement on every device, steps per launch - the same bits
    n1.restore(); n1.tune_instances_per_wave(4); n1.set_steps_per_launch(2); n1.update(nsteps)
    gd3, gv3 = n1.gather()
    assert n1.status() == 0 and (gd3 == d0).all() and (gv3 == v0).all()
    n1.close()


def test_node_level_from_a_c_program(R, tmp_path):
    """host code stays C: tests/c/node_driver.c (gcc, no HIP headers) drives the node level - create over every visible GPU,
    set state, steps, status, rkfdNodeGather - and the gathered states equal the plain copies"""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = str(tmp_path / "node_driver")
    subprocess.run(["gcc", "-O1", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "c", "node_driver.c"),
                    "-L" + os.path.join(root, "roki-fd_amd"), "-lrkfd_amd", "-Wl,-rpath," + os.path.join(root, "roki-fd_amd"), "-o", exe], check=True)
    r = subprocess.run([exe, os.path.join(root, "models"), "50", "8"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout.splitlines()
    assert out[0].startswith("devices ") and out[-1] == "identical", r.stdout


@pytest.mark.parametrize("plugin", ["mlcp", "volume"])
def test_breakable_float_joints(R, oracle_cls, plugin):
    """BREAKABLE FLOAT JOINTS on the GPU (device/rkfd_dev_brf.h; reference example/model/wall.ztk:51-53, RoKi's rk_joint_brfloat
    restated in the oracle [UNVERIFIED-DEP]): the wall of the reference's wall.ztk (thresholds 200 / 10 / 10) hit by a free box,
    16 instances.  The joints give way one after the other (a brick still attached to one that came loose moves with it; forces
    on it load the joints further down), loose bricks touch their neighbours (cells of one chain).
    MLCP: free-running for 40 steps, broken flags compared every 5 steps - identical -, states at the end.
    Volume: that plugin's friction fix-ups branch on exact ties (counted in tests/test_gpu_volume.py too), so every step
    starts from the oracle's state and the steps on which the two take different branches are counted."""
    B, nsteps = 16, 40
    vol = plugin == "volume"
    sc = R.scenarios.wall_hit(batch=B, solver=R.SOLVER_VOLUME if vol else R.SOLVER_MLCP, speed=3.0 if vol else 1.0)
    b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    b.set_state(sc["dis"], sc["vel"]); b.update_init()
    os_ = []
    for i in range(B):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init(); os_.append(o)
    assert b.get_broken().tolist() == [o.get_broken().tolist() for o in os_]
    first = b.get_broken().sum()
    if vol:
        flips = 0
        for k in range(nsteps):
            st = [o.get_state() for o in os_]
            b.set_state(np.array([x[0] for x in st]), np.array([x[1] for x in st])); b.set_broken(np.array([o.get_broken() for o in os_]))
            b.update(1)
            assert b.status() == 0
            dis, vel, acc = b.get_state(); br = b.get_broken()
            for i, o in enumerate(os_):
                o.update()
                od, ov, oa = o.get_state()
                bad = _rel(dis[i], od) > 1e-9 or _rel(vel[i], ov) > 1e-7 or br[i].tolist() != o.get_broken().tolist()
                flips += int(bad)
        assert flips <= B * nsteps // 50, flips                      # (at most 2 % of the 640 instance-steps)
        br = np.array([o.get_broken() for o in os_])
        b.set_broken(br)
    else:
        for k in range(nsteps // 5):
            b.update(5)
            assert b.status() == 0
            for o in os_:
                o.update_n(5)
            assert b.get_broken().tolist() == [o.get_broken().tolist() for o in os_], k
        br = b.get_broken()
        dis, vel, acc = b.get_state()
        for i, o in enumerate(os_):
            od, ov, oa = o.get_state()
            assert _rel(dis[i], od) < 1e-8 and _rel(vel[i], ov) < 1e-7
    assert br.sum() > first + 4 and (br[:, 1:4].sum(axis=1) >= 1).sum() >= B // 2       # joints broke DURING the run, in most instances
    assert (br[:, 0] == 0).all() and (br[:, 4:] == 0).all()                             # only links that hang on such a joint can break
    # the state crosses the boundary: a second batch started from the first one's state continues identically
    dis, vel, acc = b.get_state()
    b2 = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    b2.set_state(dis, vel); b2.set_broken(br); b2.set_contact(*b.get_contact()[:3]); b2.update_init(); b2.update(3)
    b.update_init(); b.update(3)
    assert (b2.get_broken() == b.get_broken()).all() and _rel(b2.get_state()[0], b.get_state()[0]) < 1e-12


@pytest.mark.parametrize("plugin", ["mlcp", "volume"])
def test_c_arm_wall_driver(R, oracle_cls, tmp_path, plugin):
    """examples/arm_wall.c - the shape of the reference's example/chain/arm_wall_test.c: an arm under a joint-level controller on the
    host swings its hand into a column of bricks on breakable float joints; every rkFDUpdate on the GPU, against the oracle under
    the same controller.  The second brick's joint breaks first (about step 57 under MLCP, 78 under Volume), the third's follows;
    the first brick's 200 N joint holds."""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = str(tmp_path / "arm_wall")
    subprocess.run(["gcc", "-O2", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "arm_wall.c"),
                    "-L" + os.path.join(root, "roki-fd_amd"), "-lrkfd_amd", "-Wl,-rpath," + os.path.join(root, "roki-fd_amd"), "-o", exe], check=True)
    # (under the Volume plugin the comparison ends at step 100, after both joints have broken: at step 105 - three pairs in
    #  contact, hand against two loose bricks - device and oracle take different branches of the plugin's friction fix-up from states
    #  that agree to 5e-14 (tools/brf_debug.py); the tie sensitivity of that plugin is documented in DESIGN.md, it is not the joints')
    nsteps = 140 if plugin == "mlcp" else 100
    r = subprocess.run([exe, str(nsteps), os.path.join(root, "models"), plugin], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout.splitlines()
    arm = np.array([float(x) for x in out[1].split()[1:]]); wall = np.array([float(x) for x in out[2].split()[1:]])
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_MLCP if plugin == "mlcp" else R.SOLVER_VOLUME); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    a = w.reg_file(os.path.join(M, "arm_revroot.ztk")); wl = w.reg_file(os.path.join(M, "wall.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    w.pair_chain_unreg(a)
    m = w.model.contents
    dis = np.zeros(m.ndof); dis[0] = 1.40; dis[1] = 0.12
    o = oracle_cls(w.model); o.set_state(dis, np.zeros(m.ndof)); o.update_init()
    inp = np.zeros(m.nlink); t = 0.0; tc = 0.0; target = (1.9, 0.12)
    for _ in range(nsteps):
        if tc <= t + 1e-9:
            d, v, _a = o.get_state()
            for i in range(2):
                inp[w.link_offset(a) + i] = -60.0 * (d[i] - target[i]) - 3.0 * v[i]
            o.set_motor_input(inp); tc += 0.002
        assert o.update() == 0
        t += 0.001
    od = o.get_state()[0]
    ow = w.dof_offset(wl)
    assert o.get_broken()[w.link_offset(wl):w.link_offset(wl) + 4].tolist() == [0, 0, 1, 1]      # base, brick 1 (holds), bricks 2 and 3
    assert np.abs(od[ow + 6:ow + 18]).max() > (1e-3 if plugin == "mlcp" else 1e-5)                # the loose bricks have moved
    assert np.abs(arm - od[:4]).max() < 1e-6 and np.abs(wall - od[ow:ow + 18]).max() < 1e-6


@pytest.mark.parametrize("cfg", ["config2", "config3", "config4", "config1b", "arm_press"])
def test_two_instances_per_wavefront_are_bit_identical(R, cfg):
    """RKFD_W = 2 (rkfd_devmodel.h): the world-specific kernel built with two instances per wavefront - 32 lanes each, per-lane LDS
    base and instance index, ballots and broadcasts within the half, four links of a level per sweep iteration - gives bit for bit
    the states, accelerations, contact sets and contact forces of one instance per wavefront.  Odd batch: the last wavefront's
    second half is a stand-in that stores nothing."""
    B, nsteps = 37, 12
    sc = R.scenarios.arm_press(batch=B) if cfg == "arm_press" else R.scenarios.CONFIGS[cfg](batch=B)
    out = []
    for ipw in (1, 2):
        b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
        b.set_instances_per_wave(ipw); b.specialize()
        assert b.instances_per_wave() == ipw
        b.set_state(sc["dis"], sc["vel"])
        if "motor_in" in sc:
            b.set_motor_input(sc["motor_in"])
        b.set_split(3 if ipw == 2 else 1)
        b.update_init(); b.update(nsteps)
        assert b.status() == 0
        out.append((b.get_state(), b.get_contact(), b.get_pivot()))
    for x, y in zip(out[0][0] + out[0][1] + out[0][2], out[1][0] + out[1][1] + out[1][2]):
        assert np.array_equal(x, y)


def test_two_instances_per_wavefront_need_a_small_world(R):
    sc = R.scenarios.config5(batch=2)
    b = R.Batch(sc["world"], 2, max_rigid=sc["max_rigid"])
    with pytest.raises(R.RkfdError, match="two instances per wavefront need"):
        b.set_instances_per_wave(2)
    assert b.instances_per_wave() == 1


@pytest.mark.gpu
def test_tuning_the_instances_per_wavefront_keeps_state_and_snapshot(R):
    """rkfdBatchTuneInstancesPerWave: measures both mappings from the batch's state and keeps the faster; the state, the pivots and
    an earlier snapshot are as before, and the steps taken afterwards are those of an untuned batch to the last bit.  A world that
    is not eligible for two comes back with 1 and no second time."""
    B, nsteps = 300, 10
    sc = R.scenarios.config3(batch=B)
    ref = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    ref.set_state(sc["dis"], sc["vel"]); ref.update_init(); ref.update(3)
    at3 = ref.get_state()
    ref.update(nsteps)
    want = ref.get_state() + ref.get_contact()

    b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    b.set_state(sc["dis"], sc["vel"]); b.update_init()
    b.snapshot()                              # the caller's snapshot: after rkFDUpdateInit
    b.update(3)
    chosen, ms = b.tune_instances_per_wave(5)
    assert chosen in (1, 2) and ms[0] > 0 and ms[1] > 0
    assert b.instances_per_wave() == chosen
    for x, y in zip(b.get_state(), at3):
        assert np.array_equal(x, y)
    b.update(nsteps)
    assert b.status() == 0
    for x, y in zip(b.get_state() + b.get_contact(), want):
        assert np.array_equal(x, y)
    b.restore(); b.update(3)                  # the caller's snapshot survived the measurement
    for x, y in zip(b.get_state(), at3):
        assert np.array_equal(x, y)

    sc = R.scenarios.config5(batch=4)
    b = R.Batch(sc["world"], 4, max_rigid=sc["max_rigid"])
    b.set_state(sc["dis"], sc["vel"]); b.update_init()
    chosen, ms = b.tune_instances_per_wave(2)
    assert chosen == 1 and ms[0] > 0 and ms[1] < 0 and b.instances_per_wave() == 1


def _lfoot_world(R):
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VOLUME); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    h = w.reg_file(os.path.join(M, "lfoot.ztk")); w.pair_chain_unreg(h)
    w.reg_file(os.path.join(M, "floor.ztk"))
    return w


def test_volume_plugin_guards_pairs_it_cannot_clip(R, oracle_cls):
    """a world with a shape that is NOT convex under the Volume plugin (models/lfoot.ztk: a convex foot plate under an L-shaped
    bracket; the reference's mighty.ztk in the small - VERDICT r02 missing 3): accepted; standing / sliding / rocking on the plate
    it matches the oracle, the bracket's pair with the floor being watched by the plugin's own collision test; once the bracket
    itself reaches the floor the status says so (4) instead of a wrong volume being solved"""
    w = _lfoot_world(R)
    B, N = 6, 40
    dis = np.zeros((B, 6)); vel = np.zeros((B, 6))
    dis[:, 2] = -1e-4
    dis[1:, 4] = np.linspace(0.0, 0.08, B - 1)             # small tilts about y: plate edges dig in, the bracket stays clear
    vel[:, 0] = np.linspace(0.0, 0.3, B); vel[3:, 5] = 0.5
    b = R.Batch(w, B, max_rigid=4)
    b.set_state(dis, vel); b.update_init(); b.update(N)
    assert b.status() == 0
    d, v, a = b.get_state()
    seen = 0
    for i in range(B):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); o.update_n(N)
        od, ov, oa = o.get_state()
        assert _rel(d[i], od) < 1e-8 and _rel(v[i], ov) < 1e-7, i
        seen += int(np.abs(oa[2] + 9.8) > 1e-3)            # (supported: not in free fall)
    assert seen >= B - 1
    # lying on its side (rotation of 90 degrees about y): the bracket's vertices are the lowest points
    dis2 = np.zeros((2, 6)); dis2[:, 2] = 0.049; dis2[:, 4] = np.pi / 2
    b2 = R.Batch(w, 2, max_rigid=4)
    b2.set_state(dis2, np.zeros((2, 6))); b2.update_init(); b2.update(3)
    assert b2.status() == 4
    assert "cannot clip" in R.lib().rkfdHipLastError().decode()


def test_volume_plugin_accepts_worlds_with_large_shapes(R):
    """humanoid30_shell.ztk carries 128-face shells (more faces than lanes): their pairs are guarded, the soles are solved - the
    standing humanoid runs under Volume"""
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VOLUME); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    h = w.reg_file(os.path.join(M, "humanoid30_shell.ztk")); w.pair_chain_unreg(h)
    w.reg_file(os.path.join(M, "floor.ztk"))
    sc = R.scenarios.config4_volume(batch=4)
    b = R.Batch(w, 4, max_rigid=sc["max_rigid"])
    b.set_state(sc["dis"], sc["vel"]); b.update_init(); b.update(20)
    assert b.status() == 0
    ref = R.Batch(sc["world"], 4, max_rigid=sc["max_rigid"])
    ref.set_state(sc["dis"], sc["vel"]); ref.update_init(); ref.update(20)
    for x, y in zip(b.get_state(), ref.get_state()):
        assert _rel(x, y) < 1e-9          # (the shells add mass to nothing: shapes only)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["config4", "config5"])
def test_steps_per_launch_do_not_change_results(R, cfg):
    """rkfdBatchSetStepsPerLaunch: under split launches a call of n steps goes out as rounds of launches of at most k steps - a
    launch shape, not arithmetic: 1, 5 (the default), 7 (a short last round) and n steps per launch end in the same bits"""
    B, n = 96, 23
    sc = R.scenarios.CONFIGS[cfg](batch=B)
    out = []
    for k in (1, 5, 7, n):
        b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
        b.specialize(); b.set_split(3); b.set_steps_per_launch(k)
        b.set_state(sc["dis"], sc["vel"]); b.update_init(); b.update(n)
        assert b.status() == 0
        out.append(b.get_state() + b.get_contact() + b.get_pivot())
    for o in out[1:]:
        for x, y in zip(out[0], o):
            assert np.array_equal(x, y)
