"""The oracle against the committed golden vectors (self-generated, see tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["config1", "config1b", "config2", "config3", "config4", "config5", "config1_volume", "config4_volume"]


def load(cfg):
    with open(os.path.join(GOLD, cfg + ".json")) as fp:
        return json.load(fp)


@pytest.mark.parametrize("cfg", CASES)
def test_oracle_reproduces_golden(R, oracle_cls, cfg):
    g = load(cfg)
    B = len(g["instances"])
    sc = R.scenarios.CONFIGS[cfg](batch=B)
    assert np.array_equal(np.asarray(g["dis0"]), sc["dis"])          # seeded inputs are stable
    for i, rec in enumerate(g["instances"]):
        o = oracle_cls(sc["world"].model)
        o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init()
        assert np.allclose(o.get_state()[2], rec["acc_init"], rtol=1e-11, atol=1e-11)
        n = 0
        for cp in sorted(int(k) for k in rec["steps"]):
            while n < cp:
                assert o.update() == 0; n += 1
            exp = rec["steps"][str(cp)]
            d, v, a = o.get_state()
            act, typ, ref, f = o.get_contact()
            assert np.allclose(d, exp["dis"], rtol=1e-10, atol=1e-12)
            assert np.allclose(v, exp["vel"], rtol=1e-10, atol=1e-10)
            if len(act):
                assert (act == np.asarray(exp["active"])).all()
                assert np.allclose(f, np.asarray(exp["f"]).reshape(-1, 3), rtol=1e-8, atol=1e-8)
            if "wrenches" in exp:       # Volume plugin: the 6-D wrench of every pair in volumetric contact
                got = [[p["pair"]] + p["wrench"].tolist() for p in o.volume_pairs()]
                assert len(got) == len(exp["wrenches"]) > 0
                assert np.allclose(np.asarray(got), np.asarray(exp["wrenches"]), rtol=1e-7, atol=1e-7)
