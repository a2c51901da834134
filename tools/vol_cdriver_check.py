"""how closely the C driver under rkFDSetSolver( &fd, Volume ) follows the oracle (examples/boxdrop_hardsoft.c, 2 boxes)"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rkfd_pkg
R = rkfd_pkg.load()
from oracle.pyoracle import Oracle
exe = os.path.join(tempfile.mkdtemp(), "boxdrop")
subprocess.run(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "boxdrop_hardsoft.c"),
                "-L" + os.path.join(ROOT, "roki-fd_amd"), "-lrkfd_amd", "-Wl,-rpath," + os.path.join(ROOT, "roki-fd_amd"), "-o", exe], check=True)
for nsteps in (100, 200, 400):
    out = subprocess.run([exe, "2", str(nsteps), os.path.join(ROOT, "models"), "volume"], check=True, capture_output=True, text=True).stdout
    got = np.array([[float(x) for x in l.split()[2:]] for l in out.splitlines() if l.startswith("box")])
    w = R.World(solver=R.SOLVER_VOLUME)
    w.contact_info(os.path.join(R.scenarios.MODELS, "contactinfo.ztk"))
    b0 = w.reg_file(os.path.join(R.scenarios.MODELS, "box.ztk")); w.pair_chain_unreg(b0)
    b1 = w.reg_file(os.path.join(R.scenarios.MODELS, "box.ztk")); w.pair_chain_unreg(b1)
    w.reg_file(os.path.join(R.scenarios.MODELS, "floor_hardsoft.ztk"))
    dis = np.zeros(12)
    for i in range(2):
        dis[6 * i:6 * i + 6] = [0.3 * i, 1.0 if i % 2 else -1.0, 0.1 + i * 0.05, np.deg2rad(10.0 * (i + 1)), np.deg2rad(-7.0 * (i + 1)), np.deg2rad(5.0 * (i + 1))]
    o = Oracle(w.model); o.set_state(dis, np.zeros(12)); o.update_init(); o.update_n(nsteps)
    od = o.get_state()[0].reshape(2, 6)
    print(nsteps, "steps: max |dis - oracle| =", np.abs(got - od).max(), "pairs in contact", len(o.volume_pairs()), "z", od[:, 2])
