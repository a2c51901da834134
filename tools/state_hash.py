#!/usr/bin/env python3
"""Bit-identity aid for kernel work: steps a seeded batch of each named workload on the GPU and prints a SHA-256 over the
final states, accelerations, contact sets, anchors and forces.  Run before and after a change that must not move a bit.
usage: python tools/state_hash.py [--batch 1024] [--steps 40] [--no-specialize] workload ..."""
import argparse, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rkfd_pkg
R = rkfd_pkg.load()
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--no-specialize", action="store_true")
ap.add_argument("workloads", nargs="+")
a = ap.parse_args()
def vertbox(batch, pyramid=8):
    """boxes dropped flat / tilted / sliding / spinning onto the rigid floor under the Vert plugin: apex bases, slipping and
    sticking vertices, make and break - many active-set changes per QP"""
    M = R.scenarios.MODELS
    w = R.World(solver=R.SOLVER_VERT); w.contact_info(os.path.join(M, "contactinfo.ztk"))
    w.reg_file(os.path.join(M, "box.ztk")); w.reg_file(os.path.join(M, "floor.ztk"))
    if pyramid != 8:
        w.set_pyramid(pyramid)
    rng = np.random.default_rng(11)
    dis = np.zeros((batch, 6)); vel = np.zeros((batch, 6))
    dis[:, 2] = 0.0499
    dis[1:, 3:6] = rng.uniform(-0.3, 0.3, (batch - 1, 3))
    vel[:, 0] = np.linspace(0.0, 0.4, batch); vel[2:, 3:6] = rng.uniform(-1, 1, (batch - 2, 3))
    m = w.model.contents
    for i in range(1, batch):
        dis[i, 2] -= R.scenarios.lowest_vertex_z(m, dis[i], 0) + 0.0001
    return dict(world=w, dis=dis, vel=vel, max_rigid=8)


for nm in a.workloads:
    sc = vertbox(a.batch) if nm == "vertbox" else vertbox(a.batch, 4) if nm == "vertbox4" else R.scenarios.CONFIGS[nm](batch=a.batch)
    b = R.Batch(sc["world"], a.batch, max_rigid=sc["max_rigid"])
    if not a.no_specialize and b.lds_bytes <= 64 * 1024:
        b.specialize()
    b.set_state(sc["dis"], sc["vel"]); b.update_init(); b.update(a.steps)
    st = b.status()
    h = hashlib.sha256()
    for x in b.get_state() + (b.get_contact() if b.ncand else ()):
        h.update(np.ascontiguousarray(x).tobytes())
    print("%-16s status %d  %s" % (nm, st, h.hexdigest()[:32]), flush=True)
