"""The workload configurations of BASELINE.json / SURVEY.md section 8d: worlds (models +
contact info + solver) and seeded synthetic initial states.  Shared by tests and bench.py.
Host-side set-up only; no dynamics here."""
import os

import numpy as np

from . import binding as B

MODELS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models")


def splitmix64_uniform(seed, n):
    """n doubles in [0,1) from the splitmix64 stream seeded with `seed` (instance-major draws)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _m(name):
    return os.path.join(MODELS, name)


def aa_from_rpy_deg(rx, ry, rz):
    """angle-axis vector of Rz(rz) Ry(ry) Rx(rx) (degrees) - used for the deterministic box pose."""
    rx, ry, rz = np.deg2rad([rx, ry, rz])
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    l = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    a = np.linalg.norm(l)
    th = np.arctan2(a, np.trace(R) - 1.0)
    return l * (th / a) if a > 1e-12 else np.zeros(3)


def config1(batch=1, solver=B.SOLVER_VERT):
    """box over the soft half of floor_hardsoft: ELASTIC 'soft body' contact => penalty path."""
    w = B.World(solver=solver)
    w.contact_info(_m("contactinfo.ztk"))
    w.reg_file(_m("box.ztk"))
    w.reg_file(_m("floor_hardsoft.ztk"))
    dis = np.zeros((batch, 6)); vel = np.zeros((batch, 6))
    dis[:, 0:3] = (0.0, -1.0, 0.1)
    dis[:, 3:6] = aa_from_rpy_deg(10.0, 20.0, 30.0)
    return dict(name="config1_box_soft_penalty", world=w, dis=dis, vel=vel, max_rigid=0, steps=2000)


def config1_rigid(batch=1):
    """box over the hard half: RIGID 'ground body' contact with the MLCP plugin."""
    w = B.World(solver=B.SOLVER_MLCP)
    w.contact_info(_m("contactinfo.ztk"))
    w.reg_file(_m("box.ztk"))
    w.reg_file(_m("floor_hardsoft.ztk"))
    dis = np.zeros((batch, 6)); vel = np.zeros((batch, 6))
    dis[:, 0:3] = (0.0, 1.0, 0.1)
    dis[:, 3:6] = aa_from_rpy_deg(10.0, 20.0, 30.0)
    return dict(name="config1b_box_hard_mlcp", world=w, dis=dis, vel=vel, max_rigid=8, steps=2000)


def config2(batch=4096):
    """30-link serial chain, no contact, ABA only."""
    w = B.World(solver=B.SOLVER_VERT)
    w.reg_file(_m("chain30.ztk"))
    u = splitmix64_uniform(0x5EED0002, batch * 60).reshape(batch, 60)
    dis = (u[:, :30] - 0.5) * np.pi
    vel = (u[:, 30:] - 0.5) * 2.0
    return dict(name="config2_chain30_aba", world=w, dis=dis, vel=vel, max_rigid=0, steps=1000)


def _humanoid(batch, ci_file, solver, seed, model="humanoid30.ztk"):
    w = B.World(solver=solver)
    w.contact_info(_m(ci_file))
    h = w.reg_file(_m(model))
    w.reg_file(_m("floor.ztk"))
    init = w.init_dis(h)
    n = init.shape[0]
    u = splitmix64_uniform(seed, batch * (n - 6)).reshape(batch, n - 6)
    dis = np.tile(init, (batch, 1))
    dis[:, 6:] += (u - 0.5) * 0.1
    vel = np.zeros_like(dis)
    return w, dis, vel


def config3(batch=4096, model="humanoid30.ztk"):
    """30-DoF humanoid on flat ground, Vert plugin, ELASTIC ground contact => penalty."""
    w, dis, vel = _humanoid(batch, "contact_elastic.ztk", B.SOLVER_VERT, 0x5EED0003, model)
    return dict(name="config3_humanoid_penalty", world=w, dis=dis, vel=vel, max_rigid=0, steps=1000)


def config4(batch=4096, model="humanoid30.ztk", max_rigid=16):
    """30-DoF humanoid on flat ground, MLCP plugin, RIGID ground contact."""
    w, dis, vel = _humanoid(batch, "contact_rigid.ztk", B.SOLVER_MLCP, 0x5EED0004, model)
    return dict(name="config4_humanoid_mlcp", world=w, dis=dis, vel=vel, max_rigid=max_rigid, steps=1000)


CONFIGS = {"config1": config1, "config1b": config1_rigid, "config2": config2, "config3": config3, "config4": config4}
