"""Randomised structure tests: the host-built tables (fixed-link merging, levels, slot-stable sweep schedule
with register carries, Ia pool, float slots) against the oracle on random tree topologies."""
import os

import numpy as np
import pytest

from randtree import random_tree_ztk


def _world(R, tmp_path, seed, nlink, root, shapes=0, solver=None, floor=False, motors=False):
    f = tmp_path / f"rand{seed}.ztk"
    f.write_text(random_tree_ztk(seed, nlink, root=root, shapes=shapes, motors=motors))
    w = R.World(solver=R.SOLVER_MLCP if solver is None else solver)
    if floor:
        w.contact_info(os.path.join(R.scenarios.MODELS, "contactinfo.ztk"))
    h = w.reg_file(str(f))
    if floor:
        # as the reference's drivers do for an articulated chain (example/chain/arm_box_test.c:49): the tree's OWN pairs go - random
        # joint angles put shapes of the same tree deep into each other, which is not the "falling onto the floor" this is about
        # (self-collision has its own test, with a tolerance from the measured sensitivity: test_self_collision)
        w.pair_chain_unreg(h)
        w.reg_file(os.path.join(R.scenarios.MODELS, "floor.ztk"))
    return w, h


def _state(w, seed, B):
    m = w.model.contents
    rng = np.random.default_rng(seed + 1000)
    dis = rng.uniform(-0.8, 0.8, (B, m.ndof)); vel = rng.uniform(-1.0, 1.0, (B, m.ndof))
    return dis, vel


@pytest.mark.parametrize("seed,nlink,root", [(1, 5, "float"), (2, 12, "fixed"), (3, 20, "float"), (4, 9, "revolute"),
                                              (5, 30, "float"), (6, 17, "fixed"), (7, 40, "float"), (8, 3, "revolute")])
def test_emulated_kernel_on_random_trees(R, oracle_cls, tmp_path, seed, nlink, root):
    from emu import EmuBatch
    w, _ = _world(R, tmp_path, seed, nlink, root)
    dis, vel = _state(w, seed, 2)
    eb = EmuBatch(w, 2, max_rigid=0)
    eb.set_state(dis, vel); eb.update_init(); eb.update(2)
    assert eb.status() == 0
    d, v, a = eb.get_state()
    for i in range(2):
        o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); o.update_n(2)
        od, ov, oa = o.get_state()
        for x, y in ((d[i], od), (v[i], ov), (a[i], oa)):
            assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-9, (seed, nlink, root)


@pytest.mark.gpu
def test_gpu_on_random_trees(R, oracle_cls, tmp_path):
    """60 random trees (3..48 links, all root kinds), free motion, 3 steps, vs the oracle"""
    rng = np.random.default_rng(2024)
    for k in range(60):
        seed = 100 + k
        nlink = int(rng.integers(3, 49)); root = ["float", "fixed", "revolute"][k % 3]
        w, _ = _world(R, tmp_path, seed, nlink, root)
        m = w.model.contents
        if m.ndof > 64:
            continue
        dis, vel = _state(w, seed, 4)
        b = R.Batch(w, 4, max_rigid=0)
        b.set_state(dis, vel); b.update_init(); b.update(3)
        assert b.status() == 0
        d, v, a = b.get_state()
        for i in range(4):
            o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); o.update_n(3)
            od, ov, oa = o.get_state()
            for x, y in ((d[i], od), (v[i], ov), (a[i], oa)):
                assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-8, (seed, nlink, root)


@pytest.mark.gpu
def test_gpu_on_random_trees_with_contacts(R, oracle_cls, tmp_path):
    """random float-root trees carrying box shapes fall onto the rigid floor (MLCP) and bounce: contacts
    are made and broken on links of every depth and branching; 60 steps vs the oracle"""
    rng = np.random.default_rng(77)
    ncontact = 0
    NS = 60
    for k in range(20):
        seed = 500 + k
        nlink = int(rng.integers(4, 20))
        w, h = _world(R, tmp_path, seed, nlink, "float", shapes=min(4, nlink), floor=True)
        m = w.model.contents
        B = 4
        dis = np.zeros((B, m.ndof)); vel = np.zeros((B, m.ndof))
        r2 = np.random.default_rng(seed)
        dis[:, 6:] = r2.uniform(-0.5, 0.5, (B, m.ndof - 6))
        dis[:, 3:6] = r2.uniform(-0.3, 0.3, (B, 3))
        vel[:, 2] = -0.3; vel[:, 3:6] = r2.uniform(-1.0, 1.0, (B, 3))
        for i in range(B):   # lowest collision vertex 2 mm above the floor, moving down
            dis[i, 2] -= R.scenarios.lowest_vertex_z(m, dis[i], h) - 0.002
        b = R.Batch(w, B, max_rigid=16)
        b.set_state(dis, vel); b.update_init()
        orc = []
        for i in range(B):
            o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.update_init(); orc.append(o)
        for chunk in range(NS // 10):
            b.update(10)
            assert b.status() == 0, seed
            d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
            for i, o in enumerate(orc):
                o.update_n(10)
                od, ov, oa = o.get_state(); oact, _, _, of = o.get_contact()
                assert (act[i] == oact).all(), (seed, chunk)
                # these trees are ill-conditioned for a common-origin formulation (extent ~2 m against radii of
                # gyration of ~5 cm: free-motion accelerations agree to ~3e-11, not 1e-14), and make/break events
                # amplify rounding differences further
                # (profiles/r01_parity_report.txt shows the same growth between two CPU builds of the oracle)
                tol = 1e-8 if chunk == 0 else 1e-6
                for x, y in ((d[i], od), (v[i], ov)):
                    assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < tol, (seed, nlink, chunk)
                assert np.abs(a[i] - oa).max() / max(1.0, np.abs(oa).max()) < 10 * tol, (seed, nlink, chunk)
                ncontact += int(oact.sum())
    assert ncontact > 50


@pytest.mark.gpu
def test_gpu_on_random_trees_with_motors(R, oracle_cls, tmp_path):
    """random trees whose joints carry DC motors (with stick/slip joint friction), torque motors or nothing,
    driven by random inputs (beyond the saturation limits too); 20 steps vs the oracle incl. the friction pivots"""
    rng = np.random.default_rng(9)
    for k in range(30):
        seed = 900 + k
        nlink = int(rng.integers(3, 30)); root = ["float", "fixed", "revolute"][k % 3]
        w, _ = _world(R, tmp_path, seed, nlink, root, motors=True)
        m = w.model.contents
        dis, vel = _state(w, seed, 4)
        vel *= 0.3
        inp = np.random.default_rng(seed).uniform(-30, 30, (4, m.nlink))
        b = R.Batch(w, 4, max_rigid=0)
        b.set_state(dis, vel); b.set_motor_input(inp); b.update_init(); b.update(20)
        assert b.status() == 0
        d, v, a = b.get_state(); pt, pp = b.get_pivot()
        mt = m.arr("mtype", m.nlink)
        for i in range(4):
            o = oracle_cls(w.model); o.set_state(dis[i], vel[i]); o.set_motor_input(inp[i]); o.update_init(); o.update_n(20)
            od, ov, oa = o.get_state(); opt, opp = o.get_pivot()
            for x, y in ((d[i], od), (v[i], ov), (a[i], oa)):
                assert np.abs(x - y).max() / max(1.0, np.abs(y).max()) < 1e-7, (seed, nlink, root)
            dc = mt == 2          # RKFD_MOTOR_DC
            assert (pt[i][dc] == opt[dc]).all(), (seed, nlink, root)
