"""Control for the Vert plugin's tie sensitivity (CPU only): the SAME oracle source built twice (plain -O3, and -O3 with fused
multiply-adds) runs config 4 under the Vert plugin, the second build re-synchronised to the first after every step.  Any
instance-step where the two then differ by more than 1e-8 is a QP whose active-set path depends on the last bits of its
input - the rate two correct implementations of rkfd_opt_qp.c:43-181 disagree at, against which the HIP path's rate is read.
usage: python3 tools/vert_tie_control.py [B] [nsteps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rkfd_pkg
from oracle.pyoracle import Oracle

R = rkfd_pkg.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
FMA = os.path.join(ROOT, "oracle", "_build", "librkfd_oracle_fma.so")


def rel(x, y):
    return np.abs(x - y).max() / max(1.0, np.abs(y).max())


sc = R.scenarios.config4_vert(batch=B)
bad = []
for i in range(B):
    a = Oracle(sc["world"].model); c = Oracle(sc["world"].model, FMA)
    for o in (a, c):
        o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init()
    for s in range(N):
        a.update(); c.update()
        ad, av, aa = a.get_state(); cd, cv, ca = c.get_state()
        aact, atyp, aref, af = a.get_contact(); cact, ctyp, cref, cf = c.get_contact()
        e = max(rel(cv, av), rel(cf, af))
        if e > 1e-8 or not (aact == cact).all() or not (atyp == ctyp).all():
            bad.append((i, s, e, int(aact.sum())))
        apt, app = a.get_pivot()
        c.set_state(ad, av); c.set_contact(aact, atyp, aref); c.set_pivot(apt, app)
print(f"oracle vs oracle-fma, config4v, {B} instances x {N} steps, re-synchronised every step: {len(bad)} of {B * N} instance-steps differ by more than 1e-8")
for r in bad:
    print("  instance %d step %d: %.1e (contacts %d)" % r)
