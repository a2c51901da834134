"""ZTK loader + world flattening (host C): facts about the shipped models."""
import os

import numpy as np
import pytest


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
    w.pair_chain_unreg(b2)                                          # as reference example/chain/boxdrop_test.c:37
    w.reg_file(os.path.join(R.scenarios.MODELS, "floor_hardsoft.ztk"))
    m = w.model.contents
    assert m.npair == 4                                             # 2 boxes x (ground, soft); no box-box pair
    types = m.arr("ci_type", m.nci)[m.arr("pair_ci", m.npair)]
    assert sorted(types.tolist()) == [R.CONTACT_RIGID, R.CONTACT_RIGID, R.CONTACT_ELASTIC, R.CONTACT_ELASTIC]
    # box planes: 6 unit normals
    pl = m.arr("planes", 4 * m.arr("shape_foff", m.nshape + 1)[-1]).reshape(-1, 4)
    assert np.allclose(np.linalg.norm(pl[:, :3], axis=1), 1.0)


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


REF_MODELS = "/root/reference/example/model"


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="reference checkout not present (it never is on the GPU box)")
def test_reader_loads_the_reference_shipped_models(R):
    """the independent ZTK reader against the reference's own example models (read as data): every
    model whose joints are fixed / revolute / prismatic / float loads; spherical and breakable-float
    joints are reported as unsupported, not mis-read"""
    expect = {"arm_2DoF.ztk": (3, 2), "arm_2DoF_trq.ztk": (3, 2), "box.ztk": (1, 6), "box_small.ztk": (1, 6), "crawler.ztk": (3, 6),
              "floor.ztk": (1, 0), "floor_hardsoft.ztk": (2, 0), "mighty.ztk": (25, 26), "puma.ztk": (7, 6)}
    for f, (nl, nd) in expect.items():
        w = R.World()
        w.reg_file(os.path.join(REF_MODELS, f))
        m = w.model.contents
        assert (m.nlink, m.ndof) == (nl, nd), f
    for f in ("arm.ztk", "dualarm.ztk", "wall.ztk"):
        with pytest.raises(R.RkfdError):
            R.World().reg_file(os.path.join(REF_MODELS, f))


def test_specialized_step_kernel_compiles_without_a_gpu(R):
    """rkfdSpecializeCompile: the hipRTC source of the world-specific step kernel (the world's dimensions as literals
    in front of csrc/rkfd_device.h) compiles for gfx950 - no device needed for the compile step"""
    sc = R.scenarios.config4(batch=2)
    n = R.lib().rkfdSpecializeCompile(sc["world"].model, sc["max_rigid"])
    assert n > 10000, R.lib().rkfdHipLastError().decode()
