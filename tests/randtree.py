"""Random articulated trees as ZTK text (test input generator): random topology, joint types
(revolute / prismatic / fixed, float or fixed or revolute root), inertias and frames; optionally a box
shape on some links so that contacts with a floor occur."""
import numpy as np


def _rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def random_tree_ztk(seed, nlink, root="float", shapes=0, bushy=0.5, motors=False):
    rng = np.random.default_rng(seed)
    s = f"[roki::chain]\nname : rand{seed}\n\n"
    if motors:
        s += ("[roki::motor]\nname : dcm\ntype: dc\nmotorconstant : 2.53e-2\nadmittance : 0.045872\nmaxvoltage : 24.0\nminvoltage : -24.0\n"
              "gearratio : 100.0\nrotorinertia : 2.97e-7\ngearinertia : 3.0e-7\n\n")
        s += "[roki::motor]\nname : trqm\ntype: trq\nmax : 5.0\nmin : -5.0\n\n"
    for k in range(shapes):
        s += f"[zeo::shape]\ntype : box\nname : sh{k}\ncenter : 0, 0, 0\ndepth : 0.06\nwidth : 0.05\nheight : 0.04\n\n"
    with_shape = set(rng.choice(np.arange(nlink), size=min(shapes, nlink), replace=False).tolist()) if shapes else set()
    sh = 0
    for i in range(nlink):
        if i == 0:
            jt, parent = root, None
        else:
            jt = rng.choice(["revolute", "revolute", "revolute", "prismatic", "fixed"])
            # bushy: attach to a random earlier link, else to the previous one (chain)
            parent = int(rng.integers(0, i)) if rng.random() < bushy else i - 1
        R = _rot(rng); p = rng.uniform(-0.15, 0.15, 3)
        if i == 0:
            p = np.array([0.0, 0.0, 0.5]); R = np.eye(3)
        A = rng.normal(size=(3, 3)); I = (A @ A.T) * 1e-3 + np.eye(3) * 2e-3
        com = rng.uniform(-0.03, 0.03, 3)
        s += f"[roki::link]\nname : l{i}\njointtype : {jt}\nmass : {rng.uniform(0.3, 2.0):.6f}\nstuff : body\n"
        s += f"COM : {{ {com[0]:.6f}, {com[1]:.6f}, {com[2]:.6f} }}\n"
        s += "inertia : {\n" + "".join(f" {I[r,0]:.8f}, {I[r,1]:.8f}, {I[r,2]:.8f}\n" for r in range(3)) + "}\n"
        s += "frame : {\n" + "".join(f" {R[r,0]:.10f}, {R[r,1]:.10f}, {R[r,2]:.10f}, {p[r]:.6f}\n" for r in range(3)) + "}\n"
        if motors and jt in ("revolute", "prismatic"):
            mk = rng.choice(["dcm", "trqm", ""])
            if mk:
                s += f"motor : {mk}\n"
            if mk == "dcm":
                s += f"stiffness: {rng.uniform(0, 0.5):.4f}\nviscosity: {rng.uniform(0, 0.2):.4f}\ncoulomb: {rng.uniform(0.1, 1.0):.4f}\nstaticfriction: {rng.uniform(1.0, 1.5):.4f}\n"
        if parent is not None:
            s += f"parent : l{parent}\n"
        if i in with_shape:
            s += f"shape : sh{sh}\n"; sh += 1
        s += "\n"
    return s
