"""Diagnostic (GPU box): config 4 under the Vert plugin (standing on 8 sole vertices, 64 pyramid faces) - per-step deviation of
the HIP path from the oracle, free-running and with the oracle's state re-injected after every step.
usage: python3 tools/vert_humanoid_drift.py [B] [nsteps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rkfd_pkg
from oracle.pyoracle import Oracle

R = rkfd_pkg.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60


def rel(x, y):
    return np.abs(x - y).max() / max(1.0, np.abs(y).max())


for resync in (False, True):
    sc = R.scenarios.config4_vert(batch=B)
    b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    b.set_state(sc["dis"], sc["vel"]); b.update_init()
    orc = []
    for i in range(B):
        o = Oracle(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init(); orc.append(o)
    for s in range(N):
        b.update(1)
        st = b.status()
        d, v, a = b.get_state(); act, typ, ref, f = b.get_contact()
        e = dict(dis=0.0, vel=0.0, acc=0.0, f=0.0, wrench=0.0); same = 0; nc = 0; iters = 0
        od = np.zeros_like(d); ov = np.zeros_like(v)
        oact = np.zeros_like(act); otyp = np.zeros_like(typ); oref = np.zeros_like(ref)
        optyp = []; opprev = []
        for i, o in enumerate(orc):
            o.update(); iters = max(iters, o.last_qp_iter())
            od[i], ov[i], oa = o.get_state(); oact[i], otyp[i], oref[i], of = o.get_contact()
            pt, pp = o.get_pivot(); optyp.append(pt); opprev.append(pp)
            e["dis"] = max(e["dis"], rel(d[i], od[i])); e["vel"] = max(e["vel"], rel(v[i], ov[i])); e["acc"] = max(e["acc"], rel(a[i], oa))
            e["f"] = max(e["f"], rel(f[i], of)); e["wrench"] = max(e["wrench"], rel(f[i].sum(0), of.sum(0)))
            same += int((act[i] == oact[i]).all() and (typ[i] == otyp[i] * (oact[i] != 0)).all()); nc += int(oact[i].sum())
        print("resync" if resync else "free  ", s, "status", st, "same sets/types %d/%d" % (same, B), "contacts %.1f" % (nc / B), "qp iters<=%d" % iters,
              " ".join("%s=%.1e" % kv for kv in e.items()), flush=True)
        if resync:
            b.set_state(od, ov); b.set_contact(oact, otyp, oref); b.set_pivot(np.array(optyp), np.array(opprev))
