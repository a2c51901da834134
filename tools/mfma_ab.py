"""A/B of the contact-matrix build (GPU box): VALU 3x3-block loops against the v_mfma_f64_16x16x4_f64 Gram product
(RKFD_MLCP_MFMA=1, csrc/device/rkfd_dev_mlcp.h: rkfd_mlcp_matrix_mfma).  Per variant: worst deviation from the oracle over
the rollout window, in-kernel cycles of the matrix phase (m:entries) and of the whole contact phase, and steps/s of the
specialised kernel.  usage: python3 tools/mfma_ab.py [config ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rkfd_pkg
from oracle.pyoracle import Oracle

R = rkfd_pkg.load()
NAMES = ['kin', 'cd+pen', 'sweep2', 'sweep3', 'mlcp', 'tail', 'm:entries']
for cfg in sys.argv[1:] or ["config4"]:
    sc16 = R.scenarios.CONFIGS[cfg](batch=16)
    orc = []
    for i in range(16):
        o = Oracle(sc16["world"].model); o.set_state(sc16["dis"][i], sc16["vel"][i]); o.update_init(); o.update_n(25); orc.append(o.get_state())
    sc = R.scenarios.CONFIGS[cfg](batch=4096)
    for mf in ("0", "1"):
        os.environ["RKFD_MLCP_MFMA"] = mf
        b = R.Batch(sc16["world"], 16, max_rigid=sc16["max_rigid"]); b.set_state(sc16["dis"], sc16["vel"]); b.update_init(); b.update(25)
        assert b.status() == 0
        d, v, a = b.get_state()
        err = max(np.abs(x[i] - y[i][k]).max() / max(1.0, np.abs(y[i][k]).max()) for i in range(16) for k, x in enumerate((d, v, a)) for y in (orc,))
        B = R.Batch(sc["world"], 4096, max_rigid=sc["max_rigid"]); B.set_state(sc["dis"], sc["vel"]); B.update_init(); B.update(10)
        p = B.profile(5).astype(np.float64) / 5
        ph = dict(zip(NAMES, p.mean(0)[:7]))
        B.specialize(); B.set_split(3); B.set_state(sc["dis"], sc["vel"]); B.update_init(); B.snapshot()
        for _ in range(4):
            B.restore(); B.update(25)
        B.status()
        t0 = time.perf_counter()
        for _ in range(40):
            B.restore(); B.update(25)
        B.status()
        dt = time.perf_counter() - t0
        print(f"{cfg} RKFD_MLCP_MFMA={mf}: worst deviation from the oracle after 25 steps {err:.2e};  cycles per instance-step: matrix (m:entries) {ph['m:entries']:.0f}, "
              f"contact phase {ph['mlcp']:.0f}, whole step {p[:, :6].sum(1).mean():.0f};  specialised kernel, rollouts of 25: {4096 * 1000 / dt / 1e6:.3f} M steps/s", flush=True)
