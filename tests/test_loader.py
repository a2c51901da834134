"""ZTK loader + world flattening (host C): facts about the shipped models."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_humanoid30_topology(R):
    sc = R.scenarios.config4(batch=1)
    m = sc["world"].model.contents
    assert (m.nlink, m.ndof, m.nchain) == (30, 30, 2)          # 29 humanoid links + floor
    jt = m.arr("jtype", m.nlink)
    assert (jt == R.JOINT_FLOAT).sum() == 1 and (jt == R.JOINT_REVOL).sum() == 24 and (jt == R.JOINT_FIXED).sum() == 5
    par = m.arr("parent", m.nlink)
    assert (par < np.arange(m.nlink)).all()
    assert abs(m.arr("mass", m.nlink)[:29].sum() - 5.8) < 0.3
    # two sole boxes against the floor box: 2 pairs x (8+8) candidate vertices, RIGID ground/body
    assert (m.nshape, m.npair, m.ncand) == (3, 2, 32)
    assert m.arr("ci_type", m.nci)[m.arr("pair_ci", m.npair)].tolist() == [R.CONTACT_RIGID] * 2
    assert m.solver == R.SOLVER_MLCP and m.max_iter == 10 and m.dt == 0.001
    # DC motors on every revolute joint
    assert (m.arr("mtype", m.nlink)[jt == R.JOINT_REVOL] == 2).all()


def test_chain30(R):
    sc = R.scenarios.config2(batch=1)      # keep the world alive while its model is read
    m = sc["world"].model.contents
    assert (m.nlink, m.ndof, m.ncand) == (31, 30, 0)
    org = m.arr("org", 12 * m.nlink).reshape(m.nlink, 12)
    for i in range(m.nlink):                                        # proper rotations
        Rm = org[i, :9].reshape(3, 3)
        assert np.allclose(Rm @ Rm.T, np.eye(3)) and abs(np.linalg.det(Rm) - 1) < 1e-12


def test_contact_info_association_and_pair_unreg(R):
    w = R.World(solver=R.SOLVER_MLCP)
    w.contact_info(os.path.join(R.scenarios.MODELS, "contactinfo.ztk"))
    b1 = w.reg_file(os.path.join(R.scenarios.MODELS, "box.ztk"))
    b2 = w.reg_file(os.path.join(R.scenarios.MODELS, "box.ztk"))
    w.pair_chain_unreg(b1); w.pair_chain_unreg(b2)                  # as reference example/chain/boxdrop_test.c:37: a chain's OWN pairs
    w.reg_file(os.path.join(R.scenarios.MODELS, "floor_hardsoft.ztk"))
    m = w.model.contents
    # box x box (they can land on each other: the call above drops nothing of a one-link box), 2 boxes x (ground, soft);
    # the floor's own ground - soft pair is not formed: the two links are rigidly attached to each other
    assert m.npair == 5
    assert m.arr("pair_shape", 2 * m.npair).reshape(-1, 2).tolist() == [[0, 1], [0, 2], [1, 2], [0, 3], [1, 3]]
    types = m.arr("ci_type", m.nci)[m.arr("pair_ci", m.npair)]
    assert types.tolist() == [R.CONTACT_RIGID, R.CONTACT_RIGID, R.CONTACT_RIGID, R.CONTACT_ELASTIC, R.CONTACT_ELASTIC]
    # box planes: 6 unit normals
    pl = m.arr("planes", 4 * m.arr("shape_foff", m.nshape + 1)[-1]).reshape(-1, 4)
    assert np.allclose(np.linalg.norm(pl[:, :3], axis=1), 1.0)


def test_pair_chain_unreg_drops_only_the_chains_own_pairs(R):
    """rkCDPairChainUnreg in the call order of reference example/chain/arm_box_test.c:39-49: arm, box and floor are
    registered FIRST, then the arm's pairs are unregistered - and the arm must go on touching the box and the floor.  So
    the call removes the pairs whose two cells both belong to the chain (self-collision), which registration
    (rkCDChainReg, reference src/rkfd_sim.c:198) forms by default between cells on different links of one chain."""
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_MLCP)
    w.contact_info(os.path.join(M, "contactinfo.ztk"))
    arm = w.reg_file(os.path.join(M, "arm_fold.ztk"))              # a base block and a box on each of three links
    box = w.reg_file(os.path.join(M, "box.ztk"))
    floor = w.reg_file(os.path.join(M, "floor.ztk"))

    def pairs():
        m = w.model.contents
        ch = m.arr("chain", m.nlink)[m.arr("shape_link", m.nshape)]
        ps = m.arr("pair_shape", 2 * m.npair).reshape(-1, 2)
        return [tuple(sorted((int(ch[a]), int(ch[b])))) for a, b in ps], ps
    kinds, ps = pairs()
    assert kinds.count((arm, arm)) == 6                             # 4 cells on 4 different links: every one against every other
    assert kinds.count((arm, box)) == 4 and kinds.count((arm, floor)) == 4 and kinds.count((box, floor)) == 1
    assert (ps[:, 0] < ps[:, 1]).all()                              # later cell x earlier cells, in registration order
    w.pair_chain_unreg(arm)
    kinds, _ = pairs()
    assert kinds.count((arm, arm)) == 0                             # the arm's own pairs are gone ...
    assert kinds.count((arm, box)) == 4 and kinds.count((arm, floor)) == 4 and kinds.count((box, floor)) == 1      # ... nothing else
    w.pair_chain_unreg(box)                                         # a one-link chain has no pairs of its own
    assert pairs()[0] == kinds


def test_humanoid_scenarios_carry_no_self_pairs_and_config5_has_box_box_pairs(R):
    sc = R.scenarios.config5(batch=1)
    m = sc["world"].model.contents
    ch = m.arr("chain", m.nlink)[m.arr("shape_link", m.nshape)]
    ps = m.arr("pair_shape", 2 * m.npair).reshape(-1, 2)
    kinds = [tuple(sorted((int(ch[a]), int(ch[b])))) for a, b in ps]
    assert all(a != b for a, b in kinds)                            # the humanoid's sole - sole pair was unregistered
    assert sum(1 for a, b in kinds if a < 4 and b < 4) == 6         # four boxes: six box - box pairs
    assert (m.npair, m.ncand) == (20, 320)


def test_unknown_file_fails_loudly(R):
    w = R.World()
    try:
        w.reg_file("/nonexistent/model.ztk")
    except R.RkfdError:
        return
    raise AssertionError("expected RkfdError")


def test_chain_reg_clone_unreg_in_c(R, tmp_path):
    """rkFDChainReg (clone) and rkFDChainUnreg keep the packed state and the cell offsets consistent
    (reference src/rkfd_sim.c:72-140,211-255); C program against include/roki_fd_amd.h, no GPU calls"""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = str(tmp_path / "reg_unreg")
    subprocess.run(["gcc", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "c", "reg_unreg.c"),
                    "-L" + os.path.join(root, "roki-fd_amd"), "-lrkfd_amd", "-Wl,-rpath," + os.path.join(root, "roki-fd_amd"), "-o", exe], check=True)
    out = subprocess.run([exe, os.path.join(root, "models")], check=True, capture_output=True, text=True).stdout.splitlines()
    assert out[0] == "size 36"                                            # box 6 + chain 30 + floor 0
    assert out[1] == "after clone size 42 chain ids 0 1 2 3"
    assert out[2] == "after unreg size 36 ids 0 1 2 dofoff 0 30 30  q[3]=0.25"   # the chain's joint value moved with it
    assert out[3] == "model nlink 33 ndof 36 ncand 16"                    # chain 31 + floor 1 + cloned box 1; box x floor candidates


def test_reference_api_names_in_c(R, tmp_path):
    """the names of reference include/roki_fd/rkfd_sim.h:65-67,74-80,85-93 and rkFDPrint (src/rkfd_sim.c:587-593) through
    include/roki_fd_amd.h, from C, without a GPU: another integrator than Regular + RKG is refused loudly (it
    used to be a silent no-op), rkFDSetSolver( &fd, Volume ) compiles and carries the reference's default contact info"""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = str(tmp_path / "api_surface")
    subprocess.run(["gcc", "-O1", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "c", "api_surface.c"),
                    "-L" + os.path.join(root, "roki-fd_amd"), "-lrkfd_amd", "-Wl,-rpath," + os.path.join(root, "roki-fd_amd"), "-o", exe], check=True)
    r = subprocess.run([exe, os.path.join(root, "models"), str(tmp_path / "printed.ztk")], check=True, capture_output=True, text=True)
    out = r.stdout.splitlines()
    assert out[0] == "status after Regular + RKG: 0"
    assert out[1] == "status after RK4: -2" and "no device path" in r.stderr
    assert out[2] == "time after the refused update: 0"
    assert out[3] == "status after RKG again: 0"
    assert out[4] == "Volume default relaxation 0.001"
    assert out[5] == "box shapes 1" and out[6] == "slide cell found"
    assert out[7] == "slide mode 1 vel 0.25 axis 0 0 1"
    assert out[8] == "dis[7] 0.08 vel[7] -0.16"
    assert out[9] == "read back: ok, size 4"
    assert "joint: forearm 0.125" in open(tmp_path / "printed.ztk").read() or "0.125" in open(tmp_path / "printed.ztk").read()


REF_MODELS = "/root/reference/example/model"
REF_DRIVERS = "/root/reference/example/chain"


@pytest.mark.skipif(not os.path.isdir(REF_DRIVERS), reason="reference checkout not present (it never is on the GPU box)")
def test_reference_drivers_compile_and_link_unmodified(R, tmp_path):
    """the reference's five example drivers (read where they lie, not copied) compile with -Werror=implicit-function-declaration
    against include/roki_fd/roki_fd.h and link against librkfd_amd.so as they are - #include <roki_fd/roki_fd.h>,
    rkFDODE2Assign( &fd, Regular ), rkFDSetSolver( &fd, Volume ), zVecFreeAtOnce, rkCDPairChainUnreg and all.
    (They select the Volume plugin, which has a device path for pairs of convex shapes: the worlds of the boxdrop and arm_box
    drivers are within its limits; arm_wall_test loads wall.ztk, whose breakable-float joint is refused with a message.)"""
    import glob
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    drivers = sorted(glob.glob(os.path.join(REF_DRIVERS, "*_test.c")))
    assert len(drivers) == 5
    for d in drivers:
        exe = str(tmp_path / os.path.basename(d)[:-2])
        subprocess.run(["gcc", "-O1", "-Werror=implicit-function-declaration", "-I" + os.path.join(root, "include"), d, "-L" + os.path.join(root, "roki-fd_amd"),
                        "-lrkfd_amd", "-lm", "-Wl,-rpath," + os.path.join(root, "roki-fd_amd"), "-o", exe], check=True)
        assert os.path.exists(exe)


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="reference checkout not present (it never is on the GPU box)")
def test_reader_loads_the_reference_shipped_models(R):
    """the independent ZTK reader against the reference's own example models (read as data): all 12 load - curved
    primitives (sphere / cylinder / cone) as convex polyhedra, `COM: auto` / `inertia: auto` from the shapes,
    spherical (3 DoF) and breakable-float (6 DoF) joints with their sizes and thresholds."""
    expect = {"arm_2DoF.ztk": (3, 2), "arm_2DoF_trq.ztk": (3, 2), "box.ztk": (1, 6), "box_small.ztk": (1, 6), "crawler.ztk": (3, 6),
              "floor.ztk": (1, 0), "floor_hardsoft.ztk": (2, 0), "mighty.ztk": (25, 26), "puma.ztk": (7, 6),
              "arm.ztk": (6, 12), "dualarm.ztk": None, "wall.ztk": None}
    for f, dims in expect.items():
        w = R.World()
        w.reg_file(os.path.join(REF_MODELS, f))
        m = w.model.contents
        if dims is not None:
            assert (m.nlink, m.ndof) == dims, f
        assert m.nlink > 0 and np.isfinite(m.arr("mass", m.nlink)).all() and np.isfinite(m.arr("inertia", 9 * m.nlink)).all()
    # puma.ztk: every link carries curved shapes; they arrive as convex vertex sets with face planes
    w = R.World(); w.reg_file(os.path.join(REF_MODELS, "puma.ztk")); m = w.model.contents
    assert m.nshape == 7
    voff = m.arr("shape_voff", m.nshape + 1); foff = m.arr("shape_foff", m.nshape + 1)
    assert (np.diff(voff) >= 8).all() and (np.diff(foff) >= 6).all()
    verts = m.arr("verts", 3 * voff[-1]).reshape(-1, 3); planes = m.arr("planes", 4 * foff[-1]).reshape(-1, 4)
    convex = 0
    for sh in range(m.nshape):                      # convex: every vertex on the inner side of every face plane
        v = verts[voff[sh]:voff[sh + 1]]; p = planes[foff[sh]:foff[sh + 1]]
        convex += int((v @ p[:, :3].T - p[:, 3] < 1e-9).all())
    assert convex >= 6          # the sphere, the cylinders, the cone and the forearm prism; the upper arm's loop has one reflex corner (the reader says so)
    # arm.ztk: `COM: auto` / `inertia: auto` of a link made of a sphere and a cylinder: positive definite, COM inside
    w = R.World(); w.reg_file(os.path.join(REF_MODELS, "arm.ztk")); m = w.model.contents
    I = m.arr("inertia", 9 * m.nlink).reshape(-1, 3, 3)
    for i in range(1, m.nlink):
        assert np.allclose(I[i], I[i].T) and (np.linalg.eigvalsh(I[i]) > 0).all()
    # spherical joints run on the device (three pseudo-links per joint); breakable-float joints are read, but a world
    # that holds one is refused by the device path (and the oracle) with a message
    for f in ("arm.ztk", "dualarm.ztk"):
        w = R.World(); w.pair_chain_unreg(w.reg_file(os.path.join(REF_MODELS, f)))      # (own pairs off, as the reference's arm drivers do)
        assert R.lib().rkfdLdsBytesFor(w.model, 0) > 0, R.lib().rkfdHipLastError()
    # wall.ztk: three bricks on breakable float joints - round 3: read, carried by the model (thresholds), run by device and oracle
    w = R.World(); w.reg_file(os.path.join(REF_MODELS, "wall.ztk")); m = w.model.contents
    assert (m.arr("jtype", m.nlink) == [0, 5, 5, 5]).all() and m.arr("brk_f", m.nlink).tolist() == [0.0, 200.0, 10.0, 10.0]
    assert R.lib().rkfdLdsBytesFor(w.model, 8) > 0, R.lib().rkfdHipLastError()


def test_auto_mass_properties_of_a_box(R, tmp_path):
    """`COM: auto` / `inertia: auto`: a 0.2 x 0.4 x 0.6 box of mass 3 off the origin - closed form m/12 (b^2 + c^2) ..."""
    f = tmp_path / "b.ztk"
    f.write_text("[roki::chain]\nname: b\n[zeo::shape]\nname: s\ntype: box\ncenter: 0.1, -0.2, 0.3\ndepth: 0.2\nwidth: 0.4\nheight: 0.6\n"
                 "[roki::link]\nname: l\njointtype: float\nmass: 3.0\nCOM: auto\ninertia: auto\nshape: s\n")
    w = R.World(); w.reg_file(str(f)); m = w.model.contents
    assert np.allclose(m.arr("com", 3), (0.1, -0.2, 0.3), atol=1e-14)
    assert np.allclose(m.arr("inertia", 9).reshape(3, 3), np.diag([3 / 12 * (0.16 + 0.36), 3 / 12 * (0.04 + 0.36), 3 / 12 * (0.04 + 0.16)]), atol=1e-14)


def test_malformed_polyhedron_is_refused(R, tmp_path):
    """a negative vertex index or a face naming a vertex that does not exist fails the load with a
    message instead of corrupting the heap"""
    head = "[roki::chain]\nname: b\n[zeo::shape]\nname: s\ntype: polyhedron\n"
    tail = "[roki::link]\nname: l\njointtype: float\nmass: 1\nshape: s\n"
    good = "vert: 0 { 0, 0, 0 }\nvert: 1 { 1, 0, 0 }\nvert: 2 { 0, 1, 0 }\nvert: 3 { 0, 0, 1 }\nface: 0 1 2\nface: 0 1 3\nface: 0 2 3\nface: 1 2 3\n"
    for body, ok in ((good, True), (good.replace("vert: 3", "vert: -3"), False), (good + "face: 1 2 7\n", False), (good + "face: 1 -2 3\n", False)):
        f = tmp_path / "p.ztk"; f.write_text(head + body + tail)
        if ok:
            R.World().reg_file(str(f))
        else:
            with pytest.raises(R.RkfdError):
                R.World().reg_file(str(f))


def test_ztk_writer_round_trip(R, tmp_path):
    """rkfdWorldWriteZTK (the role of rkChainFPrintZTK / rkFDPrint, reference src/rkfd_sim.c:587-593): a chain written
    and read back gives the same flattened model, bit for bit, and carries the joint displacements handed to the writer"""
    for name in ("humanoid30.ztk", "arm_revroot.ztk", "box.ztk", "floor_hardsoft.ztk"):
        w = R.World(); c = w.reg_file(os.path.join(R.scenarios.MODELS, name)); m = w.model.contents
        dis = np.linspace(0.1, 0.9, m.ndof) if m.ndof else np.zeros(1)
        out = str(tmp_path / ("w_" + name))
        assert R.lib().rkfdWorldWriteZTK(w._w, c, out.encode(), dis.ctypes.data_as(R.binding._pd)) == 0
        w2 = R.World(); c2 = w2.reg_file(out); m2 = w2.model.contents
        assert (m.nlink, m.ndof, m.nshape) == (m2.nlink, m2.ndof, m2.nshape)
        for fld, n in (("parent", m.nlink), ("jtype", m.nlink), ("mtype", m.nlink), ("org", 12 * m.nlink), ("mass", m.nlink), ("com", 3 * m.nlink),
                       ("inertia", 9 * m.nlink), ("sfric", m.nlink), ("coulomb", m.nlink), ("mot_k", m.nlink), ("mot_gear", m.nlink), ("mot_inertia", m.nlink),
                       ("shape_link", m.nshape), ("shape_voff", m.nshape + 1), ("shape_foff", m.nshape + 1)):
            assert np.array_equal(m.arr(fld, n), m2.arr(fld, n)), (name, fld)
        nv = m.arr("shape_voff", m.nshape + 1)[-1]; npl = m.arr("shape_foff", m.nshape + 1)[-1]
        assert np.array_equal(m.arr("verts", 3 * nv), m2.arr("verts", 3 * nv)) and np.allclose(m.arr("planes", 4 * npl), m2.arr("planes", 4 * npl), atol=1e-15)
        if m.ndof:
            assert np.array_equal(w2.init_dis(c2), dis)


def test_specialized_step_kernel_compiles_without_a_gpu(R):
    """rkfdSpecializeCompile: the hipRTC source of the world-specific step kernel (the world's dimensions as literals
    in front of csrc/rkfd_device.h) compiles for gfx950 - no device needed for the compile step"""
    sc = R.scenarios.config4(batch=2)
    n = R.lib().rkfdSpecializeCompile(sc["world"].model, sc["max_rigid"])
    assert n > 10000, R.lib().rkfdHipLastError().decode()


def test_specialisation_survives_the_host_changing_its_environment():
    """hipRTC lives in a private link namespace with its own copy of the C library; the host's setenv moves the environment
    array and frees the old one, which that copy still pointed at (a segmentation fault deep into a pytest process, round 2).
    rkfd_capi.hip gives the namespace a deep copy of the environment of its own before every compile (round 2 re-pointed it at
    the host's live array, which the host can move again at any time: ADVICE r02): compile, set 200 new variables, compile
    again, then change the environment once more AFTER the last compile - while threads of the namespace may still be alive -
    and exit cleanly through the interpreter's normal shutdown (exit handlers of the namespace included).  In a child
    process, so that a regression fails this test instead of killing the test run"""
    import subprocess
    import sys
    code = (
        "import os, sys, importlib\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "R = importlib.import_module('roki-fd_amd')\n"
        "sc = R.scenarios.config1(batch=2)\n"
        "n1 = R.lib().rkfdSpecializeCompile(sc['world'].model, sc['max_rigid'])\n"
        "for i in range(200):\n"
        "    os.environ['RKFD_TEST_FILLER_%d' % i] = 'x' * 100\n"
        "n2 = R.lib().rkfdSpecializeCompile(sc['world'].model, sc['max_rigid'])\n"
        "print('sizes', n1, n2)\n"
        "assert n1 > 10000 and n1 == n2\n"
        "for i in range(400):\n"
        "    os.environ['RKFD_TEST_LATE_%d' % i] = 'y' * 300\n"
        "for i in range(200):\n"
        "    del os.environ['RKFD_TEST_FILLER_%d' % i]\n"
        "print('done')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("done"), r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="reference checkout not present (it never is on the GPU box)")
def test_the_worlds_of_the_reference_drivers_build_device_tables(R):
    """the worlds the reference's five example drivers set up (example/chain/*_test.c: the models read where they lie, registered
    in the drivers' order, rkCDPairChainUnreg where the driver calls it, the plugin the driver selects): the host-side device tables
    build for all five under MLCP / Vert and under the Volume plugin, which the drivers select.  arm_wall_test.c keeps the arm's
    own pairs (no unreg call), among them motor cylinder against motor cylinder: 34 + 34 faces, where the Volume kernel takes one
    lane per face of a pair (64) - since round 3 such a pair is GUARDED (watched by the plugin's collision test, status 4 when it
    is hit) instead of the world being refused; 146 KB of LDS at the plugin's default capacity of 7 pairs, one instance per CU."""
    L = R.lib()
    M = REF_MODELS

    def world(solver, chains, unreg):
        w = R.World(solver=solver); w.contact_info(os.path.join(M, "contactinfo.ztk"))
        ids = [w.reg_file(os.path.join(M, c)) for c in chains]
        for k in unreg:
            w.pair_chain_unreg(ids[k])
        return w
    drivers = {
        "boxdrop_test": (["box.ztk"] * 3 + ["floor.ztk"], [0, 1, 2], 8),
        "boxdrop_hardsoft_test": (["box.ztk"] * 3 + ["floor_hardsoft.ztk"], [0, 1, 2], 8),
        "arm_box_test": (["arm_2DoF.ztk", "box.ztk", "floor.ztk"], [0], 6),
        "arm_box_trq_test": (["arm_2DoF_trq.ztk", "box.ztk", "floor.ztk"], [0], 6),
        "arm_wall_test": (["arm_2DoF.ztk", "wall.ztk", "floor.ztk"], [], 7),
    }
    for name, (chains, unreg, cap) in drivers.items():
        for solver in (R.SOLVER_MLCP, R.SOLVER_VERT):
            w = world(solver, chains, unreg)
            assert L.rkfdLdsBytesFor(w.model, 4 if solver == R.SOLVER_VERT else 24) > 0, (name, solver, L.rkfdHipLastError())
        w = world(R.SOLVER_VOLUME, chains, unreg)
        n = L.rkfdLdsBytesFor(w.model, cap)
        assert 0 < n <= 160 * 1024, (name, n, L.rkfdHipLastError())
