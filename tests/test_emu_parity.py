"""Kernel LOGIC check without a GPU: the device code of roki-fd_amd/csrc/rkfd_device.h run under
the 64-thread lane emulator (tests/emu) against the oracle.  The GPU tier repeats this on real
hardware through the C ABI (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from emu import EmuBatch


@pytest.mark.parametrize("cfg,B,nsteps", [("config2", 2, 2), ("config3", 2, 2), ("config4", 2, 2), ("config1b", 1, 2), ("config5", 1, 1)])
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
