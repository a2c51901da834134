"""Kernel LOGIC check without a GPU: the device code of roki-fd_amd/csrc/rkfd_device.h run under
the 64-thread lane emulator (tests/emu) against the oracle.  The GPU tier repeats this on real
hardware through the C ABI (tests/test_gpu_parity.py)."""
import os

import numpy as np
import pytest

from emu import EmuBatch


@pytest.mark.parametrize("cfg,B,nsteps", [("config2", 2, 2), ("config3", 2, 2), ("config4", 1, 2), ("config1b", 1, 2), ("config5", 1, 1)])
def test_emulated_kernel_matches_oracle(R, oracle_cls, cfg, B, nsteps):
    sc = R.scenarios.CONFIGS[cfg](batch=B)
    eb = EmuBatch(sc["world"], B, max_rigid=sc["max_rigid"])
    eb.set_state(sc["dis"], sc["vel"])
    eb.update_init()
    eb.update(nsteps)
    assert eb.status() == 0
    dis, vel, acc = eb.get_state()
    act, typ, ref, f = eb.get_contact()
    for i in range(B):
        o = oracle_cls(sc["world"].model)
        o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init()
        for _ in range(nsteps):
            o.update()
        od, ov, oa = o.get_state()
        for x, y in ((dis[i], od), (vel[i], ov), (acc[i], oa)):
            assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-9
        if eb.ncand:
            oact, otyp, oref, of = o.get_contact()
            assert (act[i] == oact).all()
            assert np.abs(f[i] - of).max() / max(1.0, np.abs(of).max()) < 1e-9
            # stick anchors: at the boundary they are in the model link's frame (floor_hardsoft's second link is merged on the device)
            assert np.abs(ref[i] - oref * (oact[:, None] != 0)).max() < 1e-9


def test_contact_capacity_overflow_is_reported(R):
    """a box lying flat on the rigid floor has 4 contact vertices; a capacity of 2 must be reported"""
    sc = R.scenarios.config1_rigid(batch=1)
    dis = sc["dis"].copy(); dis[0, 2] = 0.0499; dis[0, 3:] = 0
    eb = EmuBatch(sc["world"], 1, max_rigid=2)
    eb.set_state(dis, sc["vel"])
    eb.update_init()
    assert eb.status() == 2


@pytest.mark.parametrize("root,with_box", [("fixed", True), ("revolute", True), ("fixed", False)])
def test_emulated_arm_press_matches_oracle(R, oracle_cls, root, with_box):
    """contact paths that end at a fixed root / a 1-DoF root, a rigid pair with two moving sides, prismatic
    joint, DC and torque motors with non-zero inputs (scenarios.arm_press)"""
    B, nsteps = 2, 4
    sc = R.scenarios.arm_press(batch=B, root=root, with_box=with_box)
    eb = EmuBatch(sc["world"], B, max_rigid=sc["max_rigid"])
    eb.set_state(sc["dis"], sc["vel"]); eb.set_motor_input(sc["motor_in"])
    eb.update_init(); eb.update(nsteps)
    assert eb.status() == 0
    dis, vel, acc = eb.get_state()
    act, typ, ref, f = eb.get_contact()
    for i in range(B):
        o = oracle_cls(sc["world"].model)
        o.set_state(sc["dis"][i], sc["vel"][i]); o.set_motor_input(sc["motor_in"][i]); o.update_init()
        o.update_n(nsteps)
        od, ov, oa = o.get_state()
        for x, y in ((dis[i], od), (vel[i], ov), (acc[i], oa)):
            assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-9
        oact, otyp, oref, of = o.get_contact()
        assert (act[i] == oact).all()
        assert np.abs(f[i] - of).max() / max(1.0, np.abs(of).max()) < 1e-9


def test_emulated_vert_rigid_qp_matches_oracle(R, oracle_cls):
    """the Vert plugin's rigid branch (friction pyramids + active-set QP) under the lane emulator: a tilted box
    landing on the rigid floor (first steps run through apex-degenerate bases with >= 3 faces of a pyramid
    active) vs the oracle, whose KKT solves use a generic pseudo-inverse"""
    import os
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VERT); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "box.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    dis = np.zeros((2, 6)); vel = np.zeros((2, 6))
    dis[:, 2] = 0.0499; dis[1, 3:6] = (0.01, 0.02, 0.3); vel[:, 0] = (0.0, 0.05)
    eb = EmuBatch(w, 2, max_rigid=8)
    eb.set_state(dis, vel); eb.update_init(); eb.update(8)
    assert eb.status() == 0
    d, v, a = eb.get_state(); act, typ, ref, f = eb.get_contact()
    iters = 0
    for i in range(2):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init()
        for _ in range(8):
            o.update(); iters = max(iters, o.last_qp_iter())
        od, ov, oa = o.get_state(); oact, otyp, _, of = o.get_contact()
        assert (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all()
        for x, y in ((d[i], od), (v[i], ov), (a[i], oa)):
            assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-9
        assert np.abs(f[i] - of * (oact[:, None] != 0)).max() / max(1.0, np.abs(of).max()) < 1e-8
    assert iters >= 4          # the run did go through multi-iteration (degenerate) QPs


@pytest.mark.parametrize("solver,floor,who", [("mlcp", "floor.ztk", "box"), ("vert", "floor.ztk", "floor"), ("vert", "floor_hardsoft.ztk", "box"), ("mlcp", "floor.ztk", "both")])
def test_emulated_slide_mode_matches_oracle(R, oracle_cls, solver, floor, who):
    """cells in slide mode (fake crawler; rkFDLinkAddSlideVel, rkFDUpdateRefSlide, reference src/rkfd_util.c:26-40,218-237):
    a box whose bottom runs like a belt, a floor that does, both; rigid (MLCP / Vert QP) and elastic contacts"""
    import os
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_MLCP if solver == "mlcp" else R.SOLVER_VERT); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    bx = w.reg_file(os.path.join(M, "box.ztk")); fl = w.reg_file(os.path.join(M, floor))
    if who in ("box", "both"):
        w.set_slide(bx, 0, True, 0.2, (0.0, 1.0, 0.0))
    if who in ("floor", "both"):
        w.set_slide(fl, 0, True, -0.1, (0.3, 1.0, 0.0))
    dis = np.zeros((1, 6)); vel = np.zeros((1, 6)); dis[0, 2] = 0.0499; dis[0, 5] = 0.2
    if floor == "floor_hardsoft.ztk":
        dis[0, 1] = -1.0
    eb = EmuBatch(w, 1, max_rigid=8); eb.set_state(dis, vel); nst = 10 if solver == "vert" else 25      # (the Vert QP is slow under the emulator's barriers)
    eb.update_init(); eb.update(nst)
    assert eb.status() == 0
    o = oracle_cls(w.model); o.set_state(dis[0], vel[0]); o.update_init(); o.update_n(nst)
    d, v, a = eb.get_state(); od, ov, oa = o.get_state(); act, typ, ref, f = eb.get_contact(); oact, otyp, oref, of = o.get_contact()
    assert (act[0] == oact).all() and (typ[0] == otyp * (oact != 0)).all() and oact.sum() > 0
    assert np.abs(d[0] - od).max() < 1e-9 and np.abs(v[0] - ov).max() < 1e-9
    assert np.abs(ref[0] - oref * (oact[:, None] != 0)).max() < 1e-9
    # and the belt does move the box: a world without slide mode ends elsewhere
    w0 = R.World(solver=R.SOLVER_MLCP if solver == "mlcp" else R.SOLVER_VERT); w0.contact_info(os.path.join(M, "contactinfo.ztk"))
    w0.reg_file(os.path.join(M, "box.ztk")); w0.reg_file(os.path.join(M, floor))
    o0 = oracle_cls(w0.model); o0.set_state(dis[0], vel[0]); o0.update_init(); o0.update_n(nst)
    assert np.abs(o0.get_state()[0][:2] - od[:2]).max() > 1e-7


@pytest.mark.parametrize("which", ["ball", "shell_humanoid", "mighty"])
def test_emulated_world_with_more_than_256_candidates(R, oracle_cls, which):
    """the candidate sweep in more than four chunks of 64 (limit 1024, 16-bit contact lists, 12-bit face counts): a
    tessellated sphere on the floor (274 candidates), the humanoid with six sphere shells (764) and - where the
    reference checkout is present - the reference's UNMODIFIED mighty.ztk with all its body meshes (749 candidates: its
    non-convex shapes' vertices collide with the convex floor; the floor's vertices are not tested against them)"""
    import os
    if which == "ball":
        sc = R.scenarios.ball_roll(batch=2); nsteps = 6
    elif which == "shell_humanoid":
        sc = R.scenarios.config4_shell(batch=1); nsteps = 2
    else:
        ref = "/root/reference/example/model/mighty.ztk"
        if not os.path.exists(ref):
            pytest.skip("reference checkout not present")
        w = R.World(solver=R.SOLVER_MLCP); w.contact_info(os.path.join(R.scenarios.MODELS, "contact_rigid.ztk"))
        h = w.reg_file(ref); w.reg_file(os.path.join(R.scenarios.MODELS, "floor.ztk"))
        w.pair_chain_unreg(h)     # the robot's own pairs off, as the reference's drivers do for an articulated chain (arm_box_test.c:49)
        dis = w.init_dis(h)[None].copy()
        dis[0, 2] -= R.scenarios.lowest_vertex_z(w.model.contents, dis[0], h) + R.scenarios.SEAT_DEPTH
        sc = dict(world=w, dis=dis, vel=np.zeros_like(dis), max_rigid=8); nsteps = 2
        assert (w.model.contents.nlink, w.model.contents.ndof, w.model.contents.ncand) == (26, 26, 749)
    B = sc["dis"].shape[0]
    assert sc["world"].model.contents.ncand > 256
    eb = EmuBatch(sc["world"], B, max_rigid=sc["max_rigid"])
    eb.set_state(sc["dis"], sc["vel"]); eb.update_init(); eb.update(nsteps)
    assert eb.status() == 0
    dis, vel, acc = eb.get_state(); act, typ, ref_, f = eb.get_contact()
    seen = 0
    for i in range(B):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init()
        for _ in range(nsteps):
            o.update()
        od, ov, oa = o.get_state(); oact, otyp, oref, of = o.get_contact()
        assert (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all()
        for x, y in ((dis[i], od), (vel[i], ov), (acc[i], oa), (f[i], of)):
            assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-8
        seen += int(oact.sum())
    assert seen > 0


@pytest.mark.parametrize("which", ["free", "contact", "ref_arm"])
def test_emulated_spherical_joints(R, oracle_cls, which):
    """spherical joints: on the device a spherical joint is three pseudo-links with revolute joints about the axes of the
    joint-origin frame (three rank-1 eliminations = the rank-3 one), in the oracle a genuine 3-DoF joint with a 3x3
    joint-space inertia - two independent formulations.  models/arm_spher.ztk swinging freely and pressing its hand on
    the floor (contact paths through the pseudo-links), and - where the reference checkout is present - the
    reference's own arm.ztk (four spherical joints in series, `COM: auto` / `inertia: auto`; dualarm.ztk gives its
    links no mass - a kinematic model, it loads but has no dynamics to compare)"""
    import os
    if which in ("free", "contact"):
        sc = R.scenarios.arm_spher(batch=2, contact=which == "contact"); nsteps = 2 if which == "contact" else 5
    else:
        ref = "/root/reference/example/model/arm.ztk"
        if not os.path.exists(ref):
            pytest.skip("reference checkout not present")
        w = R.World(solver=R.SOLVER_MLCP); w.contact_info(os.path.join(R.scenarios.MODELS, "contactinfo.ztk"))
        w.pair_chain_unreg(w.reg_file(ref))           # (alone: with the floor its tessellated spheres and cylinders make 2361 candidates, above the limit of 1024)
        m = w.model.contents
        rng = np.random.default_rng(3)
        sc = dict(world=w, dis=rng.uniform(-0.4, 0.4, (2, m.ndof)), vel=rng.uniform(-1, 1, (2, m.ndof)), max_rigid=8); nsteps = 3
    B = sc["dis"].shape[0]
    eb = EmuBatch(sc["world"], B, max_rigid=sc["max_rigid"])
    eb.set_state(sc["dis"], sc["vel"]); eb.update_init(); eb.update(nsteps)
    assert eb.status() == 0
    dis, vel, acc = eb.get_state(); act, typ, ref_, f = eb.get_contact()
    seen = 0
    for i in range(B):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init()
        for _ in range(nsteps):
            o.update()
        od, ov, oa = o.get_state()
        for x, y in ((dis[i], od), (vel[i], ov), (acc[i], oa)):
            assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-8
        if eb.ncand:
            oact, otyp, oref, of = o.get_contact()
            assert (act[i] == oact).all()
            assert np.abs(f[i] - of).max() / max(1.0, np.abs(of).max()) < 1e-8
            seen += int(oact.sum())
    if which == "contact":
        assert seen > 0


def _volume_world(R, second=None):
    import os
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VOLUME)
    w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "box.ztk"))
    if second:
        w.reg_file(os.path.join(M, second))
    w.reg_file(os.path.join(M, "floor.ztk"))
    return w


def _resync_parity(eb, o, nsteps):
    """every step from the oracle's state: the worst and the median relative deviation after one rkFDUpdate"""
    errs = []
    for _ in range(nsteps):
        od, ov, _a = o.get_state()
        eb.set_state(od[None, :], ov[None, :]); eb.update_init(); eb.update(1)
        assert eb.status() == 0
        assert o.update() == 0
        d, v, a = eb.get_state(); od, ov, oa = o.get_state()
        errs.append(max(np.abs(d[0] - od).max(), np.abs(v[0] - ov).max(), np.abs(a[0] - oa).max() / max(1.0, np.abs(oa).max())))
    return max(errs), float(np.median(errs))


@pytest.mark.parametrize("case,nsteps", [("rest", 4), ("slide", 3), ("tilt", 4), ("stack", 3)])
def test_emulated_volume_plugin_matches_oracle(R, oracle_cls, case, nsteps):
    """the Volume plugin's device path (csrc/device/rkfd_dev_volume.h) against the oracle's restatement, step by step from
    the oracle's state: a box resting / sliding (kinetic-friction LP) / dropped tilted (corner and edge contacts, static-friction
    LP) on the floor (a few steps each here: the emulator's barriers make the simplex and Jacobi loops slow; the GPU tier runs
    hundreds), and a small box on the box (three rigid pairs, one between two moving bodies).  Typical deviation 1e-13;
    corner contacts with intersection volumes of 1e-10 m^3 reach 1e-7 (the 6-D QP is solved through a Cholesky factor here
    and through a pseudo-inverse of the KKT matrix there; two roundings of the oracle itself stay within 1e-11 there)."""
    w = _volume_world(R, "box_small.ztk" if case == "stack" else None)
    n = w.model.contents.ndof
    dis = np.zeros(n); vel = np.zeros(n)
    dis[2] = 0.0499
    if case == "slide":
        vel[0] = 0.5
    if case == "tilt":
        dis[:6] = R.scenarios.config1_rigid(batch=1)["dis"][0]; dis[:3] = (0, 0, 0.06)
    if case == "stack":
        dis[6:9] = (0.01, 0.02, 0.1 - 1e-4 + 0.025 - 1e-4); dis[11] = 0.4; vel[6] = 0.2
    eb = EmuBatch(w, 1, max_rigid=4)
    o = oracle_cls(w.model); o.set_state(dis, vel); o.update_init()
    if case == "tilt":
        o.update_n(361)         # down to the corner contacts (intersection volumes of 1e-9 .. 1e-11 m^3)
    if case == "stack":
        assert len(o.volume_pairs()) == 2
    worst, med = _resync_parity(eb, o, nsteps)
    assert worst < (1e-6 if case == "tilt" else 1e-9), (worst, med)


def test_emulated_volume_trajectory(R, oracle_cls):
    """free-running (no re-synchronisation): the box dropped flat from 1 mm lands the same way on both sides"""
    w = _volume_world(R)
    dis = np.zeros((1, 6)); dis[0, 2] = 0.051
    eb = EmuBatch(w, 1, max_rigid=4)
    eb.set_state(dis, np.zeros((1, 6))); eb.update_init(); eb.update(12)
    assert eb.status() == 0
    o = oracle_cls(w.model); o.set_state(dis[0], np.zeros(6)); o.update_init(); o.update_n(12)
    d, v, a = eb.get_state(); od, ov, oa = o.get_state()
    assert np.abs(d[0] - od).max() < 1e-9 and np.abs(v[0] - ov).max() < 1e-8


def test_emulated_volume_humanoid(R, oracle_cls):
    """the standing humanoid under the Volume plugin: two sole - floor pairs, probe paths through the legs to the float base"""
    sc = R.scenarios.config4_volume(batch=1)
    eb = EmuBatch(sc["world"], 1, max_rigid=sc["max_rigid"])
    eb.set_state(sc["dis"], sc["vel"]); eb.update_init(); eb.update(2)
    assert eb.status() == 0
    o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][0], sc["vel"][0]); o.update_init(); o.update_n(2)
    assert len(o.volume_pairs()) == 2
    d, v, a = eb.get_state(); od, ov, oa = o.get_state()
    assert np.abs(d[0] - od).max() < 1e-12 and np.abs(v[0] - ov).max() < 1e-10 and np.abs(a[0] - oa).max() / np.abs(oa).max() < 1e-8


@pytest.mark.skipif(not os.path.isdir("/root/reference/example/model"), reason="reference checkout not present (it never is on the GPU box)")
def test_emulated_reference_arm_box_world_under_volume(R, oracle_cls):
    """the world of the reference's arm_box_test.c as that driver sets it up - arm_2DoF.ztk (boxes as polyhedra, two cylinders,
    a DC motor), box.ztk, floor.ztk read where they lie, rkFDSetSolver( &fd, Volume ): 11 rigid pairs, cylinder - box pairs with
    40 faces; the device tables build and two steps agree with the oracle"""
    M = "/root/reference/example/model"
    w = R.World(solver=R.SOLVER_VOLUME)
    w.contact_info(os.path.join(M, "contactinfo.ztk"))
    a = w.reg_file(os.path.join(M, "arm_2DoF.ztk")); b = w.reg_file(os.path.join(M, "box.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    m = w.model.contents
    assert m.npair == 11 + 8        # the arm's five cells on three links: eight pairs of its own ...
    w.pair_chain_unreg(a)           # ... which the driver unregisters (arm_box_test.c:49); arm x box, arm x floor, box x floor stay
    m = w.model.contents
    assert m.npair == 11
    dis = np.zeros(m.ndof); off = w.dof_offset(b); dis[off:off + 3] = (0.3, 0.0, 0.05 - 1e-5)
    eb = EmuBatch(w, 1, max_rigid=6)
    eb.set_state(dis[None, :], np.zeros((1, m.ndof))); eb.update_init(); eb.update(2)
    assert eb.status() == 0
    o = oracle_cls(w.model); o.set_state(dis, np.zeros(m.ndof)); o.update_init(); o.update_n(2)
    assert len(o.volume_pairs()) >= 1
    d, v, a = eb.get_state(); od, ov, oa = o.get_state()
    assert np.abs(d[0] - od).max() < 1e-12 and np.abs(v[0] - ov).max() < 1e-10 and np.abs(a[0] - oa).max() < 1e-8


def test_emulated_volume_sixteen_conditions(R, oracle_cls):
    """a 16-sided cylinder on its end: 16 contact-plane conditions, the static-friction LP with three columns per lane"""
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VOLUME)
    w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "cylinder.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    dis = np.zeros(6); dis[3] = np.pi / 2; dis[2] = 0.06 - 1e-5
    vel = np.zeros(6); vel[0] = 0.05
    eb = EmuBatch(w, 1, max_rigid=1); eb.set_state(dis[None, :], vel[None, :]); eb.update_init(); eb.update(2)
    assert eb.status() == 0
    o = oracle_cls(w.model); o.set_state(dis, vel); o.update_init()
    assert len(o.volume_pairs()[0]["planes"]) == 16
    o.update_n(2)
    d, v, a = eb.get_state(); od, ov, oa = o.get_state()
    assert np.abs(d[0] - od).max() < 1e-12 and np.abs(v[0] - ov).max() < 1e-10 and np.abs(a[0] - oa).max() < 1e-7


def test_emulated_volume_slide_mode(R, oracle_cls):
    """cells in slide mode under the Volume plugin: the belt velocities enter the friction fix-ups (the sliding directions at the
    corners of the contact polygon), not the plugin's 6-D velocity (reference src/rkfd_util.c:83-85) - a sliding box on a running
    floor takes another path, and the device follows the oracle"""
    M = R.scenarios.MODELS
    ends = {}
    for who in ("none", "both"):
        w = R.World(solver=R.SOLVER_VOLUME); w.contact_info(os.path.join(M, "contactinfo.ztk"))
        bx = w.reg_file(os.path.join(M, "box.ztk")); fl = w.reg_file(os.path.join(M, "floor.ztk"))
        if who == "both":
            w.set_slide(bx, 0, True, 0.2, (0.0, 1.0, 0.0)); w.set_slide(fl, 0, True, -0.4, (0.3, 1.0, 0.0))
        dis = np.zeros(6); vel = np.zeros(6); dis[2] = 0.0499; dis[5] = 0.2; vel[0] = 0.3
        o = oracle_cls(w.model); o.set_state(dis, vel); o.update_init()
        if who == "both":
            eb = EmuBatch(w, 1, max_rigid=2)
            eb.set_state(dis[None, :], vel[None, :]); eb.update_init(); eb.update(3)
            assert eb.status() == 0
            o.update_n(3)
            d, v, a = eb.get_state(); od, ov, oa = o.get_state()
            assert np.abs(d[0] - od).max() < 1e-12 and np.abs(v[0] - ov).max() < 1e-10
            o.update_n(57)
        else:
            o.update_n(60)
        ends[who] = o.get_state()[0][:2].copy()
    assert np.abs(ends["none"] - ends["both"]).max() > 1e-3


def _oracle_step_from(oracle_cls, model, motor_in, st, ct, pv, dq=None):
    """one rkFDUpdate of a fresh oracle from the given state (optionally with joint displacement j nudged: dq = (j, eps))"""
    o = oracle_cls(model); o.set_motor_input(motor_in)
    d = st[0].copy()
    if dq is not None:
        d[dq[0]] += dq[1]
    o.set_state(d, st[1]); o.set_contact(*ct[:3]); o.set_pivot(*pv); o.update_init()
    o.set_contact(*ct[:3]); o.set_pivot(*pv)
    o.update()
    return o


@pytest.mark.parametrize("solver", ["mlcp", "vert"])
def test_emulated_self_collision_matches_oracle(R, oracle_cls, solver):
    """SELF-COLLISION (scenarios.arm_fold): a rigid contact between two links of ONE chain - the pairs registration forms by
    default (reference src/rkfd_sim.c:198, "self collision" branch of src/rkfd_util.c:163-170).  Both sides of the contact
    are on the same tree: the probe paths of the two sides share the joints above their common ancestor, the contact
    matrix gets the cross terms between the sides, the force is an internal force of the arm.
    The finger bounces on the beam (the last joint's inertia is tiny) and the system is stiff by construction: the oracle's
    own d(acc)/d(q) is 5e5 ... 8e7 s^-2 in contact (measured below by nudging a joint by 1e-12 rad), so two correct
    implementations whose vertex positions differ in the last bit (1e-16 m) differ by up to 1e-8 in acc.  Every step
    therefore starts from the oracle's state, and the tolerance is 1e-9 relative PLUS what a 2e-14 rad nudge of the state
    does to the oracle's own result."""
    B, nsteps = 2, 6
    sc = R.scenarios.arm_fold(batch=B, solver=R.SOLVER_MLCP if solver == "mlcp" else R.SOLVER_VERT)
    model = sc["world"].model
    eb = EmuBatch(sc["world"], B, max_rigid=sc["max_rigid"])
    eb.set_motor_input(sc["motor_in"])
    os_ = []
    for i in range(B):
        o = oracle_cls(model)
        o.set_state(sc["dis"][i], sc["vel"][i]); o.set_motor_input(sc["motor_in"][i]); o.update_init()
        os_.append(o)
    in_contact = 0
    for k in range(nsteps):
        st = [o.get_state() for o in os_]; ct = [o.get_contact() for o in os_]; pv = [o.get_pivot() for o in os_]
        eb.set_state(np.array([x[0] for x in st]), np.array([x[1] for x in st]))
        eb.set_contact(np.array([c[0] for c in ct]), np.array([c[1] for c in ct]), np.array([c[2] for c in ct]))
        eb.set_pivot(np.array([p_[0] for p_ in pv]), np.array([p_[1] for p_ in pv]))
        eb.update(1)
        assert eb.status() == 0
        dis, vel, acc = eb.get_state()
        act, typ, ref, f = eb.get_contact()
        for i, o in enumerate(os_):
            o.update()
            od, ov, oa = o.get_state()
            oact, otyp, oref, of = o.get_contact()
            in_contact += int(oact.sum() >= 1 and np.abs(of).max() > 1.0)
            # the oracle's own sensitivity of this step to its start state
            sa = sf = 0.0
            for j in range(3):
                on = _oracle_step_from(oracle_cls, model, sc["motor_in"][i], st[i], ct[i], pv[i], dq=(j, 1e-12))
                sa = max(sa, np.abs(on.get_state()[2] - oa).max() / 1e-12); sf = max(sf, np.abs(on.get_contact()[3] - of).max() / 1e-12)
            assert np.abs(dis[i] - od).max() < 1e-9 and np.abs(vel[i] - ov).max() < 1e-9 * max(1.0, np.abs(ov).max()) + 2e-14 * sa * 1e-3
            assert np.abs(acc[i] - oa).max() < 1e-9 * max(1.0, np.abs(oa).max()) + 2e-14 * sa, (k, i, sa)
            assert (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all()
            assert np.abs(f[i] - of).max() < 1e-9 * max(1.0, np.abs(of).max()) + 2e-14 * sf, (k, i, sf)
    assert in_contact >= nsteps          # the finger does press on the beam in most steps
    # with the arm's own pairs unregistered (what the reference's arm drivers do) the finger meets nothing
    sc2 = R.scenarios.arm_fold(batch=1, unreg=True)
    o = oracle_cls(sc2["world"].model)
    o.set_state(sc2["dis"][0], sc2["vel"][0]); o.set_motor_input(sc2["motor_in"][0]); o.update_init()
    assert o.get_contact()[0].sum() == 0
    o.update_n(2)
    assert o.get_contact()[0].sum() == 0


def test_emulated_breakable_float_joints(R, oracle_cls):
    """BREAKABLE FLOAT JOINTS on the device (device/rkfd_dev_brf.h) against the oracle's restatement: the wall of the reference's
    wall.ztk (three bricks, thresholds 200 / 10 / 10) hit by a slow box - rigid contact forces pass the thresholds, the joints
    break one after the other (a brick still attached to one that came loose moves with it), the loose bricks touch their
    neighbours (cells of one chain); and the cantilever whose bending moment alone breaks its first joint at rkFDUpdateInit.
    Broken flags identical at every step, states to 1e-9."""
    import os
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_MLCP); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "wall_cantilever.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    n = w.model.contents.ndof
    eb = EmuBatch(w, 1, max_rigid=8); eb.set_state(np.zeros((1, n)), np.zeros((1, n))); eb.update_init()
    o = oracle_cls(w.model); o.set_state(np.zeros(n), np.zeros(n)); o.update_init()
    assert eb.get_broken()[0].tolist() == o.get_broken().tolist() == [0, 1, 0, 0]
    eb.update(3); o.update_n(3)
    assert np.abs(eb.get_state()[2][0] - o.get_state()[2]).max() < 1e-12 and abs(o.get_state()[2][2] + 9.80665) < 1e-12

    B, nsteps = 2, 18
    sc = R.scenarios.wall_hit(batch=10)
    pick = [2, 9]                 # two instances whose joints break in stages: the second brick's at once, the third's 14 / 2 steps later
    sc["dis"] = sc["dis"][pick]; sc["vel"] = sc["vel"][pick]
    eb = EmuBatch(sc["world"], B, max_rigid=sc["max_rigid"])
    eb.set_state(sc["dis"], sc["vel"]); eb.update_init()
    os_ = []
    for i in range(B):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init(); os_.append(o)
    assert eb.get_broken().tolist() == [o.get_broken().tolist() for o in os_]
    events = 0; last = [o.get_broken().tolist() for o in os_]
    for k in range(nsteps):
        eb.update(1)
        assert eb.status() == 0
        dis, vel, acc = eb.get_state(); act, typ, ref, f = eb.get_contact(); br = eb.get_broken()
        for i, o in enumerate(os_):
            o.update()
            od, ov, oa = o.get_state(); oact, otyp, oref, of = o.get_contact()
            assert br[i].tolist() == o.get_broken().tolist(), (k, i)
            events += int(last[i] != br[i].tolist()); last[i] = br[i].tolist()
            for x, y in ((dis[i], od), (vel[i], ov), (acc[i], oa)):
                assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-9, (k, i)
            assert (act[i] == oact).all()
            assert np.abs(f[i] - of).max() / max(1.0, np.abs(of).max()) < 1e-9
    assert events >= 2 and all(sum(l) == 2 for l in last)          # the third brick's joint broke DURING the run, the first brick's held


@pytest.mark.parametrize("cfg,B,nsteps", [("config2", 3, 1), ("config3", 3, 1), ("config4", 3, 1), ("config1b", 2, 2), ("arm_press", 3, 1)])
def test_emulated_two_instances_per_wavefront(R, oracle_cls, cfg, B, nsteps):
    """RKFD_W = 2 (rkfd_devmodel.h): two instances share a wavefront, 32 lanes each, four sweep lane groups and the list
    schedule.  Results must be those of one instance per wavefront to the last bit (same operations per instance, the cross-lane
    sums in the same association), an odd batch leaves the last wavefront half empty, and the oracle agrees as before."""
    sc = R.scenarios.arm_press(batch=B) if cfg == "arm_press" else R.scenarios.CONFIGS[cfg](batch=B)
    if cfg in ("config3", "config4"):
        # the two halves of a wavefront must be free to take different branches: different contact counts per instance
        sc["dis"] = sc["dis"].copy(); sc["dis"][1::2, 2] += 0.004
    out = []
    for ipw in (1, 2):
        eb = EmuBatch(sc["world"], B, max_rigid=sc["max_rigid"], ipw=ipw)
        eb.set_state(sc["dis"], sc["vel"])
        if "motor_in" in sc:
            eb.set_motor_input(sc["motor_in"])
        eb.update_init(); eb.update(nsteps)
        assert eb.status() == 0
        out.append(eb.get_state() + eb.get_contact())
    for x, y in zip(*out):
        assert np.array_equal(x, y)
    dis, vel, acc = out[1][:3]
    for i in range(B):
        o = oracle_cls(sc["world"].model)
        o.set_state(sc["dis"][i], sc["vel"][i])
        if "motor_in" in sc:
            o.set_motor_input(sc["motor_in"][i])
        o.update_init(); o.update_n(nsteps)
        for x, y in zip((dis[i], vel[i], acc[i]), o.get_state()):
            assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-9


def test_emulated_vert_qp_wide_form(R, oracle_cls):
    """more unknowns / pyramid faces than lanes (rkfd_vert_qp_wide): config 5 under the Vert plugin - 24 contact vertices, 72
    unknowns, 192 faces - and a box world with capacity 16 x 8 faces, vs the oracle"""
    import os
    sc = R.scenarios.config5_vert(batch=1)
    eb = EmuBatch(sc["world"], 1, max_rigid=sc["max_rigid"])
    eb.set_state(sc["dis"], sc["vel"]); eb.update_init(); eb.update(2)
    assert eb.status() == 0
    d, v, a = eb.get_state(); act, typ, ref, f = eb.get_contact()
    o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][0], sc["vel"][0]); o.update_init(); o.update_n(2)
    od, ov, oa = o.get_state(); oact, otyp, _, of = o.get_contact()
    assert act[0].sum() == 24 and (act[0] == oact).all() and (typ[0] == otyp * (oact != 0)).all()
    for x, y in ((d[0], od), (v[0], ov), (a[0], oa)):
        assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-8
    assert np.abs(f[0] - of * (oact[:, None] != 0)).max() / max(1.0, np.abs(of).max()) < 1e-5

    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VERT); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "box.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    dis = np.zeros((2, 6)); vel = np.zeros((2, 6))
    dis[:, 2] = 0.0499; dis[1, 3:6] = (0.01, 0.02, 0.3); vel[:, 0] = (0.0, 0.05)
    eb = EmuBatch(w, 2, max_rigid=16)
    eb.set_state(dis, vel); eb.update_init(); eb.update(8)
    assert eb.status() == 0
    d, v, a = eb.get_state(); act, typ, ref, f = eb.get_contact()
    for i in range(2):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); o.update_n(8)
        od, ov, oa = o.get_state(); oact, otyp, _, of = o.get_contact()
        assert (act[i] == oact).all() and (typ[i] == otyp * (oact != 0)).all()
        for x, y in ((d[i], od), (v[i], ov), (a[i], oa)):
            assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-9
        assert np.abs(f[i] - of * (oact[:, None] != 0)).max() / max(1.0, np.abs(of).max()) < 1e-8


def test_emulated_volume_plugin_guards_a_shape_that_is_not_convex(R, oracle_cls):
    """models/lfoot.ztk under the Volume plugin (a convex foot plate under an L-shaped bracket): on its plate the device code
    matches the oracle and the bracket's pair with the floor is only watched; lying on its side the bracket reaches the floor:
    status 4 on the device, a counted guard hit in the oracle, no force from that pair in either"""
    import os
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VOLUME); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    h = w.reg_file(os.path.join(M, "lfoot.ztk")); w.pair_chain_unreg(h)
    w.reg_file(os.path.join(M, "floor.ztk"))
    dis = np.zeros((2, 6)); vel = np.zeros((2, 6))
    dis[:, 2] = -1e-4; dis[1, 4] = 0.05; vel[:, 0] = (0.0, 0.1)
    eb = EmuBatch(w, 2, max_rigid=4)
    eb.set_state(dis, vel); eb.update_init(); eb.update(3)
    assert eb.status() == 0
    d, v, a = eb.get_state()
    for i in range(2):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); o.update_n(3)
        assert o.volume_guard_hits() == 0
        for x, y in zip((d[i], v[i], a[i]), o.get_state()):
            assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-9
    side = np.zeros((1, 6)); side[0, 2] = 0.049; side[0, 4] = np.pi / 2
    eb = EmuBatch(w, 1, max_rigid=4)
    eb.set_state(side, np.zeros((1, 6))); eb.update_init(); eb.update(2)
    assert eb.status() == 4
    o = oracle_cls(w.model); o.set_state(side[0], np.zeros(6)); o.update_init(); o.update_n(2)
    assert o.volume_guard_hits() > 0
    for x, y in zip(eb.get_state(), o.get_state()):
        assert np.abs(x[0] - y).max() / max(1.0, np.abs(y).max()) < 1e-9
