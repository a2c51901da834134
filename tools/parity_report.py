"""Parity statement: GPU path through the C ABI vs the CPU oracle, per config, after 1, 10, 100
and N steps (default 1000): max-abs and max-rel error of acc and of the contact forces f, and how many
instances still agree to 1e-6.  Beside it the same comparison between TWO CPU builds of the oracle source
(plain -O3 and -O3 with fused multiply-adds): contact-rich runs amplify rounding differences through
stick/slip and make/break events, and that column shows how much of the long-run drift is the
system's own sensitivity rather than the GPU path.
Run on the GPU box: python3 tools/parity_report.py [nsteps] [instances]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import rkfd_pkg

R = rkfd_pkg.load()
import oracle.pyoracle as po          # noqa: E402  (checker only)
from oracle.pyoracle import Oracle     # noqa: E402

NSTEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
NINST = int(sys.argv[2]) if len(sys.argv) > 2 else 16
CHECK = sorted({1, 10, 100, NSTEPS})


def oracle_run(sc, libpath):
    """acc, f, act, dis of every instance at every checkpoint with the oracle library at libpath"""
    po._lib = None; po.LIB_PATH = libpath
    B = sc["dis"].shape[0]
    orc = []
    for i in range(B):
        o = Oracle(sc["world"].model); o.set_state(sc["dis"][i], sc["vel"][i]); o.update_init(); orc.append(o)
    out, done = {}, 0
    for upto in CHECK:
        for o in orc:
            o.update_n(upto - done)
        done = upto
        acc = np.array([o.get_state()[2] for o in orc]); dis = np.array([o.get_state()[0] for o in orc])
        if orc[0].ncand:
            cs = [o.get_contact() for o in orc]
            act = np.array([c[0] for c in cs]); f = np.array([c[3] * (c[0][:, None] != 0) for c in cs])
        else:
            act = np.zeros((B, 0), dtype=np.int32); f = np.zeros((B, 0, 3))
        out[upto] = (acc, f, act, dis)
    for o in orc:
        o.close()
    return out


def rel_per_instance(x, y):
    if x.size == 0:
        return np.zeros(x.shape[0]), np.zeros(x.shape[0])
    d = np.abs(x - y).reshape(x.shape[0], -1).max(axis=1)
    return d, d / np.maximum(1.0, np.abs(y).reshape(y.shape[0], -1).max(axis=1))


def line(tag, name, upto, a, b):
    aa, ar = rel_per_instance(a[0], b[0]); fa, fr = rel_per_instance(a[1], b[1]); _, dr = rel_per_instance(a[3], b[3])
    same = bool((a[2] == b[2]).all())
    ok = int(((ar < 1e-6) & (fr < 1e-6)).sum())
    print(f"{tag:14s} {name:9s} {upto:6d}   {aa.max():10.3e} {ar.max():10.3e}   {fa.max():10.3e} {fr.max():10.3e}   {dr.max():10.3e}   "
          f"{ok:3d}/{len(ar)}   {same}", flush=True)


subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all", "fma"], check=True, stdout=subprocess.DEVNULL)
LIB = os.path.join(ROOT, "oracle", "_build", "librkfd_oracle.so")
LIB_FMA = os.path.join(ROOT, "oracle", "_build", "librkfd_oracle_fma.so")
print(f"# {NINST} instances per config; rel = max|d| / max(1, max|oracle|) per instance, worst instance shown;")
print("# 'agree' = instances whose acc and f both agree to 1e-6; 'sets' = identical active-contact sets")
print("# pair           config     steps   acc: max-abs  max-rel    f: max-abs  max-rel    dis: max-rel   agree   sets")
for name in ("config1", "config1b", "config2", "config3", "config4", "config4v", "config5"):
    sc = R.scenarios.CONFIGS[name](batch=NINST)
    B = sc["dis"].shape[0]
    ref = oracle_run(sc, LIB)
    fma = oracle_run(sc, LIB_FMA)
    b = R.Batch(sc["world"], B, max_rigid=sc["max_rigid"])
    b.set_state(sc["dis"], sc["vel"]); b.update_init()
    done = 0
    for upto in CHECK:
        b.update(upto - done); done = upto
        assert b.status() == 0, name
        dis, vel, acc = b.get_state()
        act, typ, rf, f = b.get_contact()
        line("gpu-vs-oracle", name, upto, (acc, f, act, dis), ref[upto])
        line("oracle-vs-fma", name, upto, fma[upto], ref[upto])
