"""GPU parity INSIDE sustained rigid contact (reference src/rkfd_mlcp.c:124-284: system, PGS, SetForce): the
headline workloads of BASELINE.json with the contact problem they name - the 30-DoF humanoid standing on all 8
sole vertices (24x24 MLCP), with clutter 24 vertices (72x72) - and the rocking regime the reference's algorithm
settles into (3-5 vertices, stick/slip transitions).  The HIP path through the C ABI, generic and
world-specialised kernels, against the CPU oracle, step by step.

Tolerance.  With 8 coplanar contact vertices on one (stiction-locked) body the contact-space matrix is
rank 6 + relaxation: A = J M^-1 J' + 1e-4 I has condition number ~1e5 (measured: eigenvalues 1e-4 .. 9.4), so
formulations that agree to 1e-15 in A differ by ~1e-10 .. 1e-9 in the PGS forces and accelerations (the
emulated kernel shows 3e-10 at the very first step).  Stated tolerance for these tests: 1e-8 relative to
max(1, |oracle value|), contact sets and stick/slip types identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-8


def _relerr(x, y):
    return np.abs(x - y).max() / max(1.0, np.abs(y).max())


def _compare(b, oracles, step, min_contacts=None):
    dis, vel, acc = b.get_state()
    act, typ, ref, f = b.get_contact()
    ptyp, pprev = b.get_pivot()
    ncs = []
    for i, o in enumerate(oracles):
        od, ov, oa = o.get_state()
        oact, otyp, oref, of = o.get_contact()
        on = oact != 0
        assert (act[i] == oact).all(), f"step {step} instance {i}: contact sets differ"
        assert (typ[i] == otyp * on).all(), f"step {step} instance {i}: stick/slip types differ"
        for name, x, y in (("dis", dis[i], od), ("vel", vel[i], ov), ("acc", acc[i], oa), ("f", f[i], of)):
            e = _relerr(x, y)
            assert e < TOL, f"step {step} instance {i}: {name} differs by {e:.2e}"
        assert np.abs(ref[i] - oref * on[:, None]).max() < TOL
        optyp, opprev = o.get_pivot()
        assert (ptyp[i] == optyp).all(), f"step {step} instance {i}: joint friction pivot types differ"
        assert _relerr(pprev[i], opprev) < TOL
        ncs.append(int(on.sum()))
    if min_contacts is not None:
        assert np.mean(ncs) >= min_contacts, f"step {step}: mean contacts {np.mean(ncs)} - not the workload this test names"
    return np.mean(ncs)


@pytest.mark.parametrize("spec", [False, True], ids=["generic", "specialised"])
@pytest.mark.parametrize("cfg,B,nsteps,minc", [("config4", 16, 25, 7.0), ("config5", 8, 25, 22.0)])
def test_standing_contact_parity(R, oracle_cls, cfg, B, nsteps, minc, spec):
    """the bench's rollout window: 25 steps from the standing states, every step compared (free-running on both sides)"""
    sc = R.scenarios.CONFIGS[cfg](batch=B)
    b = R.Batch(sc["world"], B, device=0, max_rigid=sc["max_rigid"])
    if spec:
        b.specialize()
    b.set_state(sc["dis"], sc["vel"]); b.update_init()
    oracles = []
    for i in range(B):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init(); oracles.append(o)
    _compare(b, oracles, 0, minc)
    seen = []
    for s in range(1, nsteps + 1):
        b.update(1)
        assert b.status() == 0
        for o in oracles:
            assert o.update() == 0
        seen.append(_compare(b, oracles, s))
    assert np.mean(seen) >= minc, seen
    rg, el, n = b.contact_stats()
    assert n == B * nsteps and abs(rg - np.mean(seen)) < 1e-12 and el == 0      # the device's own count of what it solved


@pytest.mark.parametrize("spec", [False, True], ids=["generic", "specialised"])
@pytest.mark.parametrize("cfg,B,preroll,nsteps", [("config4", 16, 150, 40), ("config5", 6, 150, 30)])
def test_rocking_regime_parity(R, oracle_cls, cfg, B, preroll, nsteps, spec):
    """the regime the algorithm settles into after ~40 steps: 3-5 contact vertices making and breaking, stick/slip
    transitions.  The oracle is rolled `preroll` steps, its whole state (joint state, contact-vertex state, friction
    pivots) is injected through the C ABI, then both run freely and are compared at every step."""
    sc = R.scenarios.CONFIGS[cfg](batch=B)
    m = sc["world"].model.contents
    oracles = []
    dis = np.zeros((B, m.ndof)); vel = np.zeros((B, m.ndof))
    act = np.zeros((B, m.ncand), dtype=np.int32); typ = np.zeros_like(act); ref = np.zeros((B, m.ncand, 3))
    ptyp = np.zeros((B, m.nlink), dtype=np.int32); pprev = np.zeros((B, m.nlink))
    for i in range(B):
        o = oracle_cls(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init()
        assert o.update_n(preroll) == 0
        dis[i], vel[i], _ = o.get_state()
        act[i], typ[i], ref[i], _ = o.get_contact()
        ptyp[i], pprev[i] = o.get_pivot()
        oracles.append(o)
    b = R.Batch(sc["world"], B, device=0, max_rigid=sc["max_rigid"])
    if spec:
        b.specialize()
    b.set_state(dis, vel); b.set_contact(act, typ, ref); b.set_pivot(ptyp, pprev)
    changes = 0
    prev = act.copy()
    for s in range(1, nsteps + 1):
        b.update(1)
        assert b.status() == 0
        for o in oracles:
            assert o.update() == 0
        _compare(b, oracles, preroll + s)
        cur = b.get_contact()[0]
        changes += int((cur != prev).sum()); prev = cur
    assert changes > 0, "no contact was made or broken: not the regime this test names"


def test_snapshot_restore_and_rollouts(R):
    """rkfdBatchSnapshot / rkfdBatchRestore: a rollout after a restore repeats the first one bit for bit, under split
    launches too; rkfdBatchContactStats counts what the committing evaluations solved"""
    B, H = 96, 12
    sc = R.scenarios.config4(batch=B)
    for split in (1, 3):
        b = R.Batch(sc["world"], B, device=0, max_rigid=sc["max_rigid"])
        b.set_split(split)
        b.set_state(sc["dis"], sc["vel"]); b.update_init()
        assert b.status() == 0
        with pytest.raises(R.RkfdError):
            b.restore()                                   # nothing to restore yet
        b.snapshot()
        b.contact_stats(reset=True)
        b.update(H)
        first = [x.copy() for x in b.get_state()] + [x.copy() for x in b.get_contact()] + [x.copy() for x in b.get_pivot()]
        rg, el, n = b.contact_stats(reset=True)
        assert n == B * H and rg > 7.0 and el == 0
        b.update(7)                                       # wander off, then come back
        b.restore()
        start = b.get_state()
        assert np.array_equal(start[0], sc["dis"]) and np.array_equal(start[1], sc["vel"])
        b.contact_stats(reset=True)
        b.update(H)
        again = list(b.get_state()) + list(b.get_contact()) + list(b.get_pivot())
        for x, y in zip(first, again):
            assert np.array_equal(x, y)
        rg2, _, n2 = b.contact_stats()
        assert n2 == B * H and rg2 == rg
        assert b.status() == 0


def test_status_is_reported_once(R):
    """a device-side condition is reported by one rkfdBatchStatus call and cleared: the next call tells what happened
    since (the flag used to stay set for the life of the batch)"""
    sc = R.scenarios.config1_rigid(batch=1)
    dis = sc["dis"].copy(); dis[0, 2] = 0.0499; dis[0, 3:] = 0      # the box flat on the floor: 4 contact vertices
    b = R.Batch(sc["world"], 1, device=0, max_rigid=2)
    b.set_state(dis, sc["vel"]); b.update_init()
    assert b.status() == 2
    assert b"max_rigid" in R.lib().rkfdHipLastError()
    assert b.status() == 0                                           # nothing ran in between
    b.update(1)
    assert b.status() == 2                                           # and it is raised again when it happens again


@pytest.mark.parametrize("specialize", [False, True], ids=["generic", "specialised"])
def test_grouped_gauss_seidel_is_bit_identical(R, specialize, monkeypatch):
    """config 5 (humanoid + four boxes: five independent bodies, 24 contacts): the grouped Gauss-Seidel - the bodies' update
    sequences side by side in DPP rows, 8 + 8 instead of 24 + 24 updates per sweep - gives bit for bit the states and forces of
    the one-after-the-other loop (rkfdDebugVariants( 8 )); an update never touches the residuals of another body.  Both storages of
    the row blocks: sweep order (rows of at most 8 contacts, the default here) and the packed triangle (rkfdDebugVariants( 32 ): what
    longer rows use)"""
    sc = R.scenarios.config5(batch=32)
    out = []
    for mask in (0, 32, 8):      # rkfdDebugVariants: product defaults / sweep-order storage off (packed triangle) / grouped form off
        R.lib().rkfdDebugVariants(mask)
        try:
            b = R.Batch(sc["world"], 32, max_rigid=sc["max_rigid"])
        finally:
            R.lib().rkfdDebugVariants(0)
        if specialize:
            b.specialize()
        b.set_state(sc["dis"], sc["vel"]); b.update_init(); b.update(40)
        assert b.status() == 0
        out.append((b.get_state(), b.get_contact()))
    (s0, c0) = out[2]
    for s1, c1 in out[:2]:
        for x, y in zip(s1, s0):
            assert np.array_equal(x, y)
        assert np.array_equal(c1[0], c0[0]) and np.array_equal(c1[3], c0[3]) and c1[0].sum(1).min() >= 16
