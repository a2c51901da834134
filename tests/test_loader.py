"""ZTK loader + world flattening (host C): facts about the shipped models."""
import os

import numpy as np


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
