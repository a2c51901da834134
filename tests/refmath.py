"""Independent numpy mechanics used to pin the oracle (known-answer tests).

A classical (non-spatial) recursive Newton-Euler inverse dynamics in world coordinates,
written from the textbook vector equations; it shares no code and no formulation with
oracle/rkfd_oracle.c (link-frame articulated-body algorithm) or the device code
(world-frame spatial ABA).  Joint conventions are those of include/rkfd_model.h.
"""
import numpy as np

G = 9.80665
FIXED, REVOL, PRISM, FLOAT = 0, 1, 2, 3


def rot_aa(aa):
    th = np.linalg.norm(aa)
    if th < 1e-12:
        return np.eye(3)
    k = aa / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def model_arrays(m):
    """numpy views of the ctypes RkfdModel"""
    nl = m.nlink
    d = dict(nlink=nl, ndof=m.ndof,
             parent=m.arr("parent", nl), jtype=m.arr("jtype", nl), dofoff=m.arr("dofoff", nl),
             org=m.arr("org", 12 * nl).reshape(nl, 12), mass=m.arr("mass", nl),
             com=m.arr("com", 3 * nl).reshape(nl, 3), inertia=m.arr("inertia", 9 * nl).reshape(nl, 3, 3),
             mtype=m.arr("mtype", nl), mot_gear=m.arr("mot_gear", nl), mot_inertia=m.arr("mot_inertia", nl),
             mot_k=m.arr("mot_k", nl), mot_admit=m.arr("mot_admit", nl))
    return d


def fk(md, q):
    """world frames (R, p) of every link and the world orientation of each joint-origin frame"""
    nl = md["nlink"]
    R = np.zeros((nl, 3, 3)); p = np.zeros((nl, 3)); Row = np.zeros((nl, 3, 3))
    for i in range(nl):
        Ro = md["org"][i, :9].reshape(3, 3); po = md["org"][i, 9:]
        par = md["parent"][i]
        Rp, pp = (np.eye(3), np.zeros(3)) if par < 0 else (R[par], p[par])
        off = md["dofoff"][i]; jt = md["jtype"][i]
        Row[i] = Rp @ Ro
        if jt == REVOL:
            c, s = np.cos(q[off]), np.sin(q[off])
            R[i] = Rp @ Ro @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]); p[i] = pp + Rp @ po
        elif jt == PRISM:
            R[i] = Rp @ Ro; p[i] = pp + Rp @ (po + Ro[:, 2] * q[off])
        elif jt == FLOAT:
            R[i] = Rp @ Ro @ rot_aa(q[off + 3:off + 6]); p[i] = pp + Rp @ (po + Ro @ q[off:off + 3])
        else:
            R[i] = Rp @ Ro; p[i] = pp + Rp @ po
    return R, p, Row


def rnea(md, q, qd, qdd, fext=None, gravity=G):
    """generalized forces that produce qdd at (q, qd).  fext: dict link -> list of (point_w, force_w).
    Float joints return (force, moment about the link origin) projected on the joint-origin axes."""
    nl = md["nlink"]
    R, p, Row = fk(md, q)
    w = np.zeros((nl, 3)); al = np.zeros((nl, 3)); a = np.zeros((nl, 3))
    for i in range(nl):
        par = md["parent"][i]; off = md["dofoff"][i]; jt = md["jtype"][i]
        if par < 0:
            wp = np.zeros(3); alp = np.zeros(3); ap = np.zeros(3); pp = np.zeros(3)
        else:
            wp, alp, ap, pp = w[par], al[par], a[par], p[par]
        r = p[i] - pp
        a[i] = ap + np.cross(alp, r) + np.cross(wp, np.cross(wp, r))
        w[i] = wp; al[i] = alp
        z = R[i][:, 2]
        if jt == REVOL:
            w[i] = wp + z * qd[off]
            al[i] = alp + z * qdd[off] + np.cross(wp, z * qd[off])
        elif jt == PRISM:
            a[i] += 2 * np.cross(wp, z * qd[off]) + z * qdd[off]
        elif jt == FLOAT:
            vj = Row[i] @ qd[off:off + 3]; wj = Row[i] @ qd[off + 3:off + 6]
            w[i] = wp + wj
            al[i] = alp + Row[i] @ qdd[off + 3:off + 6] + np.cross(wp, wj)
            a[i] += 2 * np.cross(wp, vj) + Row[i] @ qdd[off:off + 3]
    f = np.zeros((nl, 3)); n = np.zeros((nl, 3))   # force / moment about the link origin, world frame
    for i in range(nl):
        cw = R[i] @ md["com"][i]
        Iw = R[i] @ md["inertia"][i] @ R[i].T
        ac = a[i] + np.cross(al[i], cw) + np.cross(w[i], np.cross(w[i], cw))
        F = md["mass"][i] * (ac - np.array([0, 0, -gravity]))
        N = Iw @ al[i] + np.cross(w[i], Iw @ w[i])
        f[i] = F; n[i] = N + np.cross(cw, F)
        if fext and i in fext:
            for (x, fw) in fext[i]:
                f[i] -= fw; n[i] -= np.cross(np.asarray(x) - p[i], fw)
    tau = np.zeros(md["ndof"])
    for i in range(nl - 1, -1, -1):
        par = md["parent"][i]; off = md["dofoff"][i]; jt = md["jtype"][i]
        z = R[i][:, 2]
        if jt == REVOL:
            tau[off] = z @ n[i]
        elif jt == PRISM:
            tau[off] = z @ f[i]
        elif jt == FLOAT:
            tau[off:off + 3] = Row[i].T @ f[i]; tau[off + 3:off + 6] = Row[i].T @ n[i]
        if par >= 0:
            f[par] += f[i]; n[par] += n[i] + np.cross(p[i] - p[par], f[i])
    return tau


def mass_matrix(md, q):
    n = md["ndof"]
    z = np.zeros(n)
    h0 = rnea(md, q, z, z, gravity=0.0)
    M = np.zeros((n, n))
    for k in range(n):
        e = np.zeros(n); e[k] = 1
        M[:, k] = rnea(md, q, z, e, gravity=0.0) - h0
    return M


def point_jacobian_T(md, q, link, x, dirs):
    """columns J' d for world directions d of a force applied at world point x on `link`
    (generalized force of the external force), via inverse dynamics"""
    n = md["ndof"]
    z = np.zeros(n)
    h0 = rnea(md, q, z, z, gravity=0.0)
    cols = []
    for d in dirs:
        h = rnea(md, q, z, z, fext={link: [(x, np.asarray(d, dtype=float))]}, gravity=0.0)
        cols.append(h0 - h)
    return np.array(cols).T
