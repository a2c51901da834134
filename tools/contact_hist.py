"""Contact history of the standing workloads under the reference's algorithm (CPU oracle; DESIGN.md "Scenario note"): mean number
of rigid contact vertices per 5-step window after a standing start, for several seat depths, and the force / velocity chatter of
one instance.  usage: python3 tools/contact_hist.py [instances] [steps]  ->  profiles/r02_contact_history.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rkfd_pkg
from oracle.pyoracle import Oracle

R = rkfd_pkg.load()
S = R.scenarios
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60


def history(cfg, depth):
    S.SEAT_DEPTH = depth
    sc = S.CONFIGS[cfg](batch=B)
    out = np.zeros((B, N), dtype=int)
    for i in range(B):
        o = Oracle(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init()
        for s in range(N):
            o.update(); out[i, s] = int(o.get_contact()[0].sum())
    return out


default = S.SEAT_DEPTH
print(f"mean rigid contact vertices per 5-step window, {B} instances, standing start (flat soles)")
for depth in (1e-4, 1e-5, 1e-6, 1e-7, 1e-9):
    h = history("config4", depth)
    print(f"config4  seat depth {depth:7.0e}: " + " ".join("%.2f" % x for x in h.reshape(B, N // 5, 5).mean(axis=(0, 2))))
h = history("config5", default)
print(f"config5  seat depth {default:7.0e}: " + " ".join("%.2f" % x for x in h.reshape(B, N // 5, 5).mean(axis=(0, 2))))
S.SEAT_DEPTH = default
sc = S.config4(batch=1)
o = Oracle(sc["world"].model); o.set_state(sc["dis"][0], sc["vel"][0]); o.update_init()
print("\nconfig4 instance 0, seat depth %.0e: step, contacts, base vertical velocity, base angular velocity, sum of normal forces" % default)
for s in range(48):
    o.update(); act, typ, ref, f = o.get_contact(); d, v, a = o.get_state()
    print("%3d  %d  vz %+.2e  w (%+.4f %+.4f %+.4f)  sum fz %.1f N" % (s, act.sum(), v[2], v[3], v[4], v[5], f[:, 2].sum()))
