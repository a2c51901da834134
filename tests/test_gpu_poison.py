"""LDS is not cleared between kernels: a read of storage nobody wrote takes whatever the previous kernel on that CU left behind, and
a result that depends on it passes or fails with the history of the box (seen in round 2: the grouped Gauss-Seidel multiplied zero
increments with contact-matrix entries outside the rows' blocks, which are never formed - harmless after a run of our own kernels,
NaN after somebody else's).  RKFD_DEBUG_POISON_LDS=1 makes every instance fill its LDS with all ones (NaN as a double, -1 as an
int) before it starts; the results must not change.  The emulator (tests/emu) poisons its LDS the same way on the CPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORKLOADS = ["config1", "config1b", "config2", "config3", "config4", "config4v", "config5", "config1_volume", "config4_volume", "config4_shell",
             "config4_26", "arm_press", "arm_press_revroot", "arm_press_vert", "arm_press_volume", "ball_roll", "arm_spher"]


def _scenario(R, cfg, B):
    S = R.scenarios
    if cfg == "arm_press":
        return S.arm_press(batch=B)                                   # a rigid pair with two moving sides, motors, a prismatic joint
    if cfg == "arm_press_revroot":
        return S.arm_press(batch=B, root="rev", with_box=False)
    if cfg == "arm_press_vert":
        sc = S.arm_press(batch=B, solver=R.SOLVER_VERT)
        sc["max_rigid"] = 8                                           # (64 pyramid faces: one constraint per lane)
        return sc
    if cfg == "arm_press_volume":
        return S.arm_press(batch=B, solver=R.SOLVER_VOLUME)
    if cfg == "ball_roll":
        return S.ball_roll(batch=B)                                   # 274 candidate vertices: the collision sweep in chunks
    if cfg == "arm_spher":
        return S.arm_spher(batch=B, contact=True)                     # spherical joints (three device links each)
    return S.CONFIGS[cfg](batch=B)


def _steps(R, sc, B, nsteps, specialize):
    b = R.Batch(sc["world"], B, device=0, max_rigid=sc["max_rigid"])
    if specialize:
        b.specialize()
    b.set_state(sc["dis"][:B], sc["vel"][:B])
    if "motor_in" in sc:
        b.set_motor_input(sc["motor_in"][:B])
    b.update_init(); b.update(nsteps)
    assert b.status() == 0
    return b.get_state(), b.get_contact()


@pytest.mark.parametrize("specialize", [False, True], ids=["generic", "specialised"])
@pytest.mark.parametrize("cfg", WORKLOADS)
def test_results_do_not_depend_on_what_lds_held_before(R, cfg, specialize, monkeypatch):
    B, nsteps = (256, 12) if cfg.startswith("config") else (32, 40)
    sc = _scenario(R, cfg, B)
    monkeypatch.delenv("RKFD_DEBUG_POISON_LDS", raising=False)
    (d0, v0, a0), (act0, typ0, ref0, f0) = _steps(R, sc, B, nsteps, specialize)
    monkeypatch.setenv("RKFD_DEBUG_POISON_LDS", "1")
    (d1, v1, a1), (act1, typ1, ref1, f1) = _steps(R, sc, B, nsteps, specialize)
    for x in (d1, v1, a1, f1):
        assert np.isfinite(x).all()
    assert np.array_equal(d0, d1) and np.array_equal(v0, v1) and np.array_equal(a0, a1)
    assert np.array_equal(act0, act1) and np.array_equal(typ0, typ1) and np.array_equal(f0, f1)


def test_packed_triangle_storage_of_the_grouped_solve(R, monkeypatch):
    """config 5 with the row blocks in the packed triangle (rkfdDebugVariants( 32 ): the storage rows of more than 8 contacts use,
    and the one that had the bug described above)"""
    B, nsteps = 256, 12
    sc = R.scenarios.CONFIGS["config5"](batch=B)
    R.lib().rkfdDebugVariants(32)
    try:
        monkeypatch.delenv("RKFD_DEBUG_POISON_LDS", raising=False)
        (d0, v0, a0), (act0, typ0, ref0, f0) = _steps(R, sc, B, nsteps, False)
        monkeypatch.setenv("RKFD_DEBUG_POISON_LDS", "1")
        (d1, v1, a1), (act1, typ1, ref1, f1) = _steps(R, sc, B, nsteps, False)
    finally:
        R.lib().rkfdDebugVariants(0)
    assert np.isfinite(a1).all() and np.isfinite(f1).all()
    assert np.array_equal(d0, d1) and np.array_equal(v0, v1) and np.array_equal(a0, a1) and np.array_equal(f0, f1)
