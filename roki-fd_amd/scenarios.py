"""The workload configurations of BASELINE.json / SURVEY.md section 8d: worlds (models +
contact info + solver) and seeded synthetic initial states.  Shared by tests and bench.py.
Host-side set-up only; no dynamics here."""
import os

import numpy as np

from . import binding as B

MODELS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models")


def splitmix64_uniform(seed, n, start=0):
    """draws start .. start+n-1 (doubles in [0,1)) of the splitmix64 stream seeded with `seed`.  The stream is
    index-addressable, so a rank generates only the draws of its own shard of instances (instance-major draws)."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
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


def config1(batch=1, solver=B.SOLVER_VERT, first=0):
    """box over the soft half of floor_hardsoft: ELASTIC 'soft body' contact => penalty path."""
    w = B.World(solver=solver)
    w.contact_info(_m("contactinfo.ztk"))
    w.reg_file(_m("box.ztk"))
    w.reg_file(_m("floor_hardsoft.ztk"))
    dis = np.zeros((batch, 6)); vel = np.zeros((batch, 6))
    dis[:, 0:3] = (0.0, -1.0, 0.1)
    dis[:, 3:6] = aa_from_rpy_deg(10.0, 20.0, 30.0)
    return dict(name="config1_box_soft_penalty", world=w, dis=dis, vel=vel, max_rigid=0, steps=2000)


def config1_rigid(batch=1, first=0):
    """box over the hard half: RIGID 'ground body' contact with the MLCP plugin."""
    w = B.World(solver=B.SOLVER_MLCP)
    w.contact_info(_m("contactinfo.ztk"))
    w.reg_file(_m("box.ztk"))
    w.reg_file(_m("floor_hardsoft.ztk"))
    dis = np.zeros((batch, 6)); vel = np.zeros((batch, 6))
    dis[:, 0:3] = (0.0, 1.0, 0.1)
    dis[:, 3:6] = aa_from_rpy_deg(10.0, 20.0, 30.0)
    return dict(name="config1b_box_hard_mlcp", world=w, dis=dis, vel=vel, max_rigid=8, steps=2000)


def config2(batch=4096, first=0):
    """30-link serial chain, no contact, ABA only."""
    w = B.World(solver=B.SOLVER_VERT)
    w.reg_file(_m("chain30.ztk"))
    u = splitmix64_uniform(0x5EED0002, batch * 60, start=first * 60).reshape(batch, 60)
    dis = (u[:, :30] - 0.5) * np.pi
    vel = (u[:, 30:] - 0.5) * 2.0
    return dict(name="config2_chain30_aba", world=w, dis=dis, vel=vel, max_rigid=0, steps=1000)


def _rot_aa_batch(aa):
    """rotation matrices [B,3,3] of the angle-axis vectors aa [B,3]"""
    th = np.linalg.norm(aa, axis=1)
    small = th < 1e-12
    k = aa / np.where(small, 1.0, th)[:, None]
    K = np.zeros((aa.shape[0], 3, 3))
    K[:, 0, 1] = -k[:, 2]; K[:, 0, 2] = k[:, 1]; K[:, 1, 0] = k[:, 2]
    K[:, 1, 2] = -k[:, 0]; K[:, 2, 0] = -k[:, 1]; K[:, 2, 1] = k[:, 0]
    R = np.eye(3)[None] + np.sin(th)[:, None, None] * K + (1 - np.cos(th))[:, None, None] * (K @ K)
    R[small] = np.eye(3)
    return R


def link_frames(m, Q):
    """world frames (R [B,nl,3,3], p [B,nl,3]) of every link at the joint displacements Q [B,ndof]: plain forward
    kinematics on the flattened model, vectorised over the instances (scenario set-up only)"""
    nl = m.nlink; Bn = Q.shape[0]
    parent, jt, off = m.arr("parent", nl), m.arr("jtype", nl), m.arr("dofoff", nl)
    org = m.arr("org", 12 * nl).reshape(nl, 12)
    R = np.zeros((Bn, nl, 3, 3)); p = np.zeros((Bn, nl, 3))
    eye = np.broadcast_to(np.eye(3), (Bn, 3, 3)); zero = np.zeros((Bn, 3))
    for i in range(nl):
        Ro = org[i, :9].reshape(3, 3); po = org[i, 9:]
        Rp, pp = (eye, zero) if parent[i] < 0 else (R[:, parent[i]], p[:, parent[i]])
        if jt[i] == B.JOINT_REVOL:
            c, s_ = np.cos(Q[:, off[i]]), np.sin(Q[:, off[i]])
            Rz = np.zeros((Bn, 3, 3)); Rz[:, 0, 0] = c; Rz[:, 0, 1] = -s_; Rz[:, 1, 0] = s_; Rz[:, 1, 1] = c; Rz[:, 2, 2] = 1
            R[:, i] = Rp @ Ro @ Rz; p[:, i] = pp + Rp @ po
        elif jt[i] == B.JOINT_PRISM:
            R[:, i] = Rp @ Ro; p[:, i] = pp + np.einsum("bij,bj->bi", Rp, po + Ro[:, 2][None] * Q[:, off[i]][:, None])
        elif jt[i] == B.JOINT_FLOAT:
            R[:, i] = Rp @ Ro @ _rot_aa_batch(Q[:, off[i] + 3:off[i] + 6])
            p[:, i] = pp + np.einsum("bij,bj->bi", Rp, po + Q[:, off[i]:off[i] + 3] @ Ro.T)
        else:
            R[:, i] = Rp @ Ro; p[:, i] = pp + Rp @ po
    return R, p


def chain_vertices_batch(m, Q, chain):
    """world positions [B,nv,3] of the collision vertices of `chain` (shape order, vertex order) at Q [B,ndof]"""
    nl = m.nlink
    R, p = link_frames(m, Q)
    ch = m.arr("chain", nl)
    voff = m.arr("shape_voff", m.nshape + 1); slink = m.arr("shape_link", m.nshape)
    verts = m.arr("verts", 3 * voff[-1]).reshape(-1, 3)
    out = [np.zeros((Q.shape[0], 0, 3))]
    for sh in range(m.nshape):
        l = slink[sh]
        if ch[l] == chain:
            out.append(p[:, l][:, None, :] + np.einsum("bij,vj->bvi", R[:, l], verts[voff[sh]:voff[sh + 1]]))
    return np.concatenate(out, axis=1)


def chain_vertices(m, q, chain):
    """the same for one instance: [nv,3]"""
    return chain_vertices_batch(m, np.asarray(q, dtype=np.float64)[None], chain)[0]


def lowest_vertex_z(m, q, chain):
    """height of the lowest collision vertex of `chain` at joint displacement q"""
    return chain_vertices(m, q, chain)[:, 2].min()


# Depth at which resting vertices are seated in the floor.  A rigid contact's compensation term asks for the
# separating velocity K*depth (reference src/rkfd_mlcp.c:178-186; K = 1000 in contactinfo.ztk): it must stay
# below what gravity takes back in a step, g*dt = 9.8e-3 m/s, or the body is launched off the floor
# (round 1 seated at 0.5 mm = 0.5 m/s and measured ballistic flight).
SEAT_DEPTH = 1.0e-5


def seat_soles_flat(m, dis, chain, base_off, nominal, depth=None, iters=6):
    """Stand the chain on flat soles: Gauss-Newton on (base height, base roll, base pitch, the two revolute
    joints above each sole shape) until the lowest four vertices of every SOLE of `chain` (a collision shape that reaches
    the floor at the nominal pose) sit
    `depth` in the floor z = 0.  dis [B,ndof] is modified in place (all other coordinates keep their values); the
    "lowest four" are picked once, at the pose `nominal` [ndof], so that every shard of a batch picks the same.
    Vectorised over the instances; a handful of forward-kinematics passes."""
    depth = SEAT_DEPTH if depth is None else depth
    nl = m.nlink
    parent, jt, off, ch = m.arr("parent", nl), m.arr("jtype", nl), m.arr("dofoff", nl), m.arr("chain", nl)
    slink = m.arr("shape_link", m.nshape); voff = m.arr("shape_voff", m.nshape + 1)
    unknown = [base_off + 2, base_off + 3, base_off + 4]
    sel = []; nv = 0
    v0 = chain_vertices(m, nominal, chain)
    zmin = v0[:, 2].min()
    for sh in range(m.nshape):
        l = slink[sh]
        if ch[l] != chain:
            continue
        n = int(voff[sh + 1] - voff[sh])
        if v0[nv:nv + n, 2].min() > zmin + 1e-3:       # not a sole: a shape that does not reach the floor at the nominal pose
            nv += n
            continue
        k = 0
        while l >= 0 and k < 2:
            if jt[l] == B.JOINT_REVOL:
                unknown.append(int(off[l])); k += 1
            l = parent[l]
        n = int(voff[sh + 1] - voff[sh])
        sel.extend(nv + np.argsort(v0[nv:nv + n, 2], kind="stable")[:4]); nv += n
    sel = np.array(sel); unknown = np.array(unknown)
    eps = 1e-7
    for _ in range(iters):
        r = chain_vertices_batch(m, dis, chain)[:, sel, 2] + depth
        J = np.zeros(r.shape + (len(unknown),))
        for c, u in enumerate(unknown):
            d2 = dis.copy(); d2[:, u] += eps
            J[:, :, c] = (chain_vertices_batch(m, d2, chain)[:, sel, 2] + depth - r) / eps
        dis[:, unknown] -= np.einsum("bij,bj->bi", np.linalg.pinv(J, rcond=1e-10), r)
    return dis


def _humanoid(batch, ci_file, solver, seed, model="humanoid30.ztk", first=0):
    """standing pose of [roki::chain::init] + per-instance joint perturbation U(-0.05,0.05) rad,
    then stood on flat soles (seat_soles_flat): with the perturbed ankles the soles are tilted by up to 0.05 rad,
    the robot touches the floor with ONE vertex and ends rocking on three - not the "standing, Nc ~ 8" workload
    the configs name.  Instances first .. first+batch-1 of the seeded stream (a rank builds only its shard)."""
    w = B.World(solver=solver)
    w.contact_info(_m(ci_file))
    h = w.reg_file(_m(model))
    w.reg_file(_m("floor.ztk"))
    w.pair_chain_unreg(h)              # no self-collision pairs, as the reference's drivers ask for an articulated chain (arm_box_test.c:49)
    init = w.init_dis(h)
    n = init.shape[0]
    u = splitmix64_uniform(seed, batch * (n - 6), start=first * (n - 6)).reshape(batch, n - 6)
    dis = np.tile(init, (batch, 1))
    dis[:, 6:] += (u - 0.5) * 0.1
    seat_soles_flat(w.model.contents, dis, h, w.dof_offset(h), nominal=init)
    vel = np.zeros_like(dis)
    return w, dis, vel


def config3(batch=4096, model="humanoid30.ztk", first=0):
    """30-DoF humanoid on flat ground, Vert plugin, ELASTIC ground contact => penalty."""
    w, dis, vel = _humanoid(batch, "contact_elastic.ztk", B.SOLVER_VERT, 0x5EED0003, model, first)
    return dict(name="config3_humanoid_penalty", world=w, dis=dis, vel=vel, max_rigid=0, steps=1000)


def config4(batch=4096, model="humanoid30.ztk", max_rigid=8, first=0):
    """30-DoF humanoid on flat ground, MLCP plugin, RIGID ground contact."""
    w, dis, vel = _humanoid(batch, "contact_rigid.ztk", B.SOLVER_MLCP, 0x5EED0004, model, first)
    return dict(name="config4_humanoid_mlcp", world=w, dis=dis, vel=vel, max_rigid=max_rigid, steps=1000)


def config4_vert(batch=4096, model="humanoid30.ztk", first=0):
    """config 4 under the reference's DEFAULT plugin: 30-DoF humanoid on flat ground, RIGID ground contact,
    Vert plugin (8-face friction pyramids + active-set QP).  Capacity 8 contact vertices = 64 pyramid faces,
    one per lane."""
    w, dis, vel = _humanoid(batch, "contact_rigid.ztk", B.SOLVER_VERT, 0x5EED0004, model, first)
    return dict(name="config4_humanoid_vert_qp", world=w, dis=dis, vel=vel, max_rigid=8, steps=1000)


def config5(batch=4096, max_rigid=24, first=0, solver=None):
    """config 4 + clutter: four small boxes resting on the floor around the feet; box-floor, box-foot and box-box
    pairs are RIGID ('ground body' / 'body body' of contactinfo.ztk).  Registration and rkCDPairChainUnreg in the order
    of the reference's box-drop drivers (boxdrop_test.c:27-39: the call drops a chain's OWN pairs - none for a one-link box;
    the humanoid's sole-sole pair goes).  54 joint coordinates, 34 links, 320 candidate contact vertices per instance
    (6 box-box + 8 box-sole + 4 box-floor + 2 sole-floor pairs x 16).  Contact capacity 24 vertices = 72 MLCP rows (two
    rows per lane).  solver = SOLVER_VERT: the same world under the reference's default plugin - 72 unknowns and 192 pyramid
    faces, the wide form of the QP (rkfd_vert_qp_wide)."""
    w = B.World(solver=B.SOLVER_MLCP if solver is None else solver)
    w.contact_info(_m("contactinfo.ztk"))
    boxes = []
    for _ in range(4):
        c = w.reg_file(_m("box_small.ztk"))
        w.pair_chain_unreg(c)          # as reference example/chain/boxdrop_test.c:37
        boxes.append(c)
    h = w.reg_file(_m("humanoid30.ztk"))
    w.pair_chain_unreg(h)
    w.reg_file(_m("floor.ztk"))
    init = w.init_dis(h)
    n = init.shape[0]
    u = splitmix64_uniform(0x5EED0005, batch * (n - 6), start=first * (n - 6)).reshape(batch, n - 6)
    m = w.model.contents
    dis = np.zeros((batch, m.ndof))
    ho = w.dof_offset(h)
    dis[:, ho:ho + n] = init
    dis[:, ho + 6:ho + n] += (u - 0.5) * 0.1
    for k, (sx, sy) in enumerate(((1, 1), (-1, 1), (-1, -1), (1, -1))):
        o = w.dof_offset(boxes[k])
        dis[:, o:o + 3] = (0.15 * sx, 0.15 * sy, 0.025 - SEAT_DEPTH)
    nominal = np.zeros(m.ndof); nominal[ho:ho + n] = init
    seat_soles_flat(m, dis, h, ho, nominal=nominal)
    vel = np.zeros_like(dis)
    return dict(name="config5_humanoid_clutter_mlcp" if solver is None else "config5_humanoid_clutter_vert_qp", world=w, dis=dis, vel=vel, max_rigid=max_rigid, steps=1000)


def arm_press(batch=8, root="fixed", with_box=True, seed=0x5EED00A1, solver=B.SOLVER_MLCP):
    """TEST scenario (not one of BASELINE's configs): a 4-joint arm (yaw with DC motor and joint friction,
    shoulder pitch with torque motor, telescopic forearm, wrist roll) presses its box-shaped hand onto a
    free box lying on the rigid floor (or onto the floor itself), MLCP plugin.  Contact paths end at a
    fixed root / at a 1-DoF root, one rigid pair has two moving sides, joints are prismatic as well as
    revolute, motor inputs are non-zero - the branches the humanoid workloads do not take."""
    w = B.World(solver=solver)
    w.contact_info(_m("contactinfo.ztk"))
    a = w.reg_file(_m("arm_fixedroot.ztk" if root == "fixed" else "arm_revroot.ztk"))
    bx = w.reg_file(_m("box.ztk")) if with_box else None
    w.reg_file(_m("floor.ztk"))
    m = w.model.contents
    u = splitmix64_uniform(seed, batch * 8).reshape(batch, 8)
    dis = np.zeros((batch, m.ndof)); vel = np.zeros((batch, m.ndof))
    inp = np.zeros((batch, m.nlink))
    ao = w.dof_offset(a)
    top = 0.1 if with_box else 0.0                       # what the hand rests on
    for b in range(batch):
        q = np.zeros(m.ndof)
        q[ao + 0] = (u[b, 0] - 0.5) * 0.6                # yaw
        q[ao + 2] = u[b, 1] * 0.05                       # forearm extension
        q[ao + 3] = (u[b, 2] - 0.5) * 0.4                # wrist roll
        lo, hi = 0.0, 1.2                                # shoulder pitch by bisection: lowest hand vertex SEAT_DEPTH inside
        for _ in range(60):
            q[ao + 1] = 0.5 * (lo + hi)
            if lowest_vertex_z(m, q, a) > top - SEAT_DEPTH:
                lo = q[ao + 1]
            else:
                hi = q[ao + 1]
        if with_box:
            v = chain_vertices(m, q, a)
            c = v[np.argsort(v[:, 2])[:4]].mean(axis=0)  # under the lowest face of the hand
            o = w.dof_offset(bx)
            q[o:o + 3] = (c[0], c[1], 0.05 - SEAT_DEPTH)
        dis[b] = q
        vel[b, ao:ao + 4] = (u[b, 3:7] - 0.5) * 0.2
        inp[b, w.link_offset(a) + (1 if root == "fixed" else 0)] = (u[b, 7] - 0.5) * 20.0     # yaw motor voltage
        inp[b, w.link_offset(a) + (2 if root == "fixed" else 1)] = 2.0                         # shoulder torque, pressing down
    return dict(name=f"arm_press_{root}{'_box' if with_box else ''}", world=w, dis=dis, vel=vel, motor_in=inp, max_rigid=12, steps=200)


def arm_fold(batch=8, seed=0x5EED00E1, solver=B.SOLVER_MLCP, unreg=False):
    """TEST scenario (SELF-COLLISION): models/arm_fold.ztk folded so that its last link's box presses on its first link's
    box - a rigid contact whose two sides are links of ONE chain (the pairs registration forms by default between the
    cells of a chain, reference src/rkfd_sim.c:198; probed through the "self collision" branch of src/rkfd_util.c:163-170).
    The joint-3 motor presses the finger down, so the contact force is an internal force of the arm.  unreg=True calls
    rkCDPairChainUnreg for the arm as the reference's arm drivers do (example/chain/arm_box_test.c:49): the finger then
    passes through the beam."""
    w = B.World(solver=solver)
    w.contact_info(_m("contactinfo.ztk"))
    a = w.reg_file(_m("arm_fold.ztk"))
    w.reg_file(_m("floor.ztk"))
    if unreg:
        w.pair_chain_unreg(a)
    m = w.model.contents
    u = splitmix64_uniform(seed, batch * 8).reshape(batch, 8)
    dis = np.zeros((batch, m.ndof)); vel = np.zeros((batch, m.ndof)); inp = np.zeros((batch, m.nlink))
    ao, lo_ = w.dof_offset(a), w.link_offset(a)
    slink = m.arr("shape_link", m.nshape); voff = m.arr("shape_voff", m.nshape + 1)
    fsh = [sh for sh in range(m.nshape) if slink[sh] == lo_ + 3][0]        # the finger's shape; vertices of the chain come in shape order
    v0 = int(voff[fsh] - voff[[sh for sh in range(m.nshape) if slink[sh] == lo_][0]])
    nv = int(voff[fsh + 1] - voff[fsh])
    for b in range(batch):
        q = np.zeros(m.ndof)
        q[ao + 0] = (u[b, 0] - 0.5) * 0.2
        q[ao + 1] = np.deg2rad(130.0 + 6.0 * (u[b, 1] - 0.5))
        lo, hi = np.deg2rad(150.0), np.deg2rad(178.0)        # joint 3: the finger's lowest vertex SEAT_DEPTH inside the beam's top face
        for _ in range(60):
            q[ao + 2] = 0.5 * (lo + hi)
            Rl, pl = link_frames(m, q[None])
            f = chain_vertices(m, q, a)[v0:v0 + nv]
            y = ((f - pl[0, lo_ + 1]) @ Rl[0, lo_ + 1])[:, 1].min() - 0.02
            if y > -SEAT_DEPTH:
                hi = q[ao + 2]
            else:
                lo = q[ao + 2]
        dis[b] = q
        vel[b, ao:ao + 3] = (u[b, 2:5] - 0.5) * 0.1
        inp[b, lo_ + 3] = -1.0 - u[b, 5]                      # joint 3 presses the finger onto the beam
        inp[b, lo_ + 1] = 2.0 * u[b, 6]                       # joint 1 carries part of the arm's weight
    return dict(name="arm_fold" + ("_unreg" if unreg else ""), world=w, dis=dis, vel=vel, motor_in=inp, max_rigid=8, steps=200)


def wall_hit(batch=4, seed=0x5EED00F1, solver=B.SOLVER_MLCP, speed=1.0):
    """TEST scenario (BREAKABLE FLOAT JOINTS): models/wall.ztk - the structure and thresholds of the reference's wall.ztk: a base
    fixed to the world and three bricks, each on a breakable float joint (200 N / N m, then 10 and 10) - and a free box flying
    at the column's upper bricks: the rigid contact forces of the impact pass the thresholds, the joints break in the order the
    forces dictate, the bricks come loose, tumble, and collide with each other (cells of ONE chain: the wall's own pairs stay
    registered, as in reference example/chain/arm_wall_test.c, which makes no rkCDPairChainUnreg call for the wall)."""
    w = B.World(solver=solver)
    w.contact_info(_m("contactinfo.ztk"))
    wl = w.reg_file(_m("wall.ztk"))
    bx = w.reg_file(_m("box.ztk"))
    w.reg_file(_m("floor.ztk"))
    m = w.model.contents
    u = splitmix64_uniform(seed, batch * 6).reshape(batch, 6)
    dis = np.zeros((batch, m.ndof)); vel = np.zeros((batch, m.ndof))
    o = w.dof_offset(bx)
    dis[:, o + 0] = -0.0997                                  # 0.2 mm in front of the bricks' face at x = -0.0495
    dis[:, o + 1] = 0.55 + 0.01 * (u[:, 0] - 0.5)
    dis[:, o + 2] = 0.22 + 0.16 * u[:, 1]                    # at the height of the second / third brick
    dis[:, o + 5] = 0.1 * (u[:, 2] - 0.5)
    # slow enough that the joints break one after the other (the 200 N joint of the first brick holds below ~0.5 m/s): a brick
    # still attached to one that has come loose moves with it, and the forces on it load the joints further down
    vel[:, o + 0] = speed * (0.05 + 0.35 * u[:, 3])
    vel[:, o + 2] = 0.1 * (u[:, 4] - 0.5)
    return dict(name="wall_hit", world=w, dis=dis, vel=vel, max_rigid=6 if solver == B.SOLVER_VOLUME else 16, steps=200)


def config5_vert(batch=4096, max_rigid=24, first=0):
    """config 5 under the reference's default plugin (Vert): 24 contact vertices = 72 unknowns, 192 pyramid faces"""
    return config5(batch=batch, max_rigid=max_rigid, first=first, solver=B.SOLVER_VERT)


def config3_26(batch=4096, first=0):
    """config 3 on the 26-DoF model with mighty.ztk's own topology (reported alongside)"""
    d = config3(batch, model="humanoid26.ztk", first=first); d["name"] = "config3_humanoid26_penalty"; return d


def config4_26(batch=4096, first=0):
    """config 4 on the 26-DoF model with mighty.ztk's own topology"""
    d = config4(batch, model="humanoid26.ztk", first=first); d["name"] = "config4_humanoid26_mlcp"; return d


def config4_shell(batch=4096, first=0):
    """config 4 on humanoid30_shell.ztk: the same robot and states with a tessellated sphere on six links - 764 candidate
    contact vertices per instance (the reference's mighty.ztk with its body meshes has 749): what the multi-chunk
    collision sweep costs"""
    d = config4(batch, model="humanoid30_shell.ztk", first=first); d["name"] = "config4_humanoid30_shell_mlcp"; return d


def ball_roll(batch=4, seed=0x5EED00B1):
    """TEST scenario: a free tessellated sphere (models/ball.ztk, 266 vertices: a world with more than 256 candidates)
    set on the rigid floor on its lowest vertex with sliding velocity and spin, MLCP plugin: it slides, sticks and rolls
    from vertex to vertex"""
    w = B.World(solver=B.SOLVER_MLCP)
    w.contact_info(_m("contactinfo.ztk"))
    b = w.reg_file(_m("ball.ztk"))
    w.reg_file(_m("floor.ztk"))
    m = w.model.contents
    u = splitmix64_uniform(seed, batch * 8).reshape(batch, 8)
    dis = np.zeros((batch, 6)); vel = np.zeros((batch, 6))
    dis[:, 3:6] = (u[:, 0:3] - 0.5) * 0.8
    dis[0, 3:6] = 0
    dis[:, 2] = 0.1
    dis[:, 2] -= chain_vertices_batch(m, dis, b)[:, :, 2].min(axis=1) + SEAT_DEPTH
    vel[:, 0:2] = (u[:, 3:5] - 0.5) * 0.6
    vel[:, 3:6] = (u[:, 5:8] - 0.5) * 8.0
    return dict(name="ball_roll", world=w, dis=dis, vel=vel, max_rigid=8, steps=200)


def arm_spher(batch=4, contact=False, seed=0x5EED00C1):
    """TEST scenario: models/arm_spher.ztk - a fixed base and three links on SPHERICAL joints (the joint type of the
    reference's arm.ztk / dualarm.ztk) over the rigid floor, MLCP plugin.  contact=False: random configurations and
    rates, swinging freely (frictionless: energy is conserved); contact=True: the arm points straight down with its
    box-shaped hand flat on the floor, the two outer joints moving"""
    w = B.World(solver=B.SOLVER_MLCP)
    w.contact_info(_m("contactinfo.ztk"))
    w.reg_file(_m("arm_spher.ztk"))
    w.reg_file(_m("floor.ztk"))
    m = w.model.contents
    u = splitmix64_uniform(seed, batch * 2 * m.ndof).reshape(batch, 2 * m.ndof)
    if contact:
        dis = np.zeros((batch, m.ndof)); vel = np.zeros((batch, m.ndof))
        dis[:, 0] = np.pi
        vel[:, 3:] = (u[:, :6] - 0.5)
    else:
        dis = (u[:, :m.ndof] - 0.5) * 1.2; vel = (u[:, m.ndof:] - 0.5) * 4.0
    return dict(name="arm_spher" + ("_contact" if contact else ""), world=w, dis=dis, vel=vel, max_rigid=8, steps=200)


def config1_volume(batch=4096, first=0, seed=0x5EED00D1):
    """the reference's own boxdrop scene under the plugin its drivers select (rkFDSetSolver( &fd, Volume )): a box lying on
    the rigid half of floor_hardsoft (seated SEAT_DEPTH deep, random yaw and position), pushed sideways - one rigid pair in
    volumetric contact in every evaluation: the 6-D wrench QP, then the static or the kinetic friction LP"""
    w = B.World(solver=B.SOLVER_VOLUME)
    w.contact_info(_m("contactinfo.ztk"))
    w.reg_file(_m("box.ztk"))
    w.reg_file(_m("floor_hardsoft.ztk"))
    u = splitmix64_uniform(seed, 6 * batch, start=6 * first).reshape(batch, 6)
    dis = np.zeros((batch, 6)); vel = np.zeros((batch, 6))
    dis[:, 0] = -0.5 + u[:, 0]; dis[:, 1] = 0.6 + 0.8 * u[:, 1]; dis[:, 2] = 0.05 - SEAT_DEPTH
    dis[:, 5] = np.pi * (u[:, 2] - 0.5)
    vel[:, 0] = 0.4 * (u[:, 3] - 0.5); vel[:, 1] = 0.4 * (u[:, 4] - 0.5); vel[:, 5] = 2.0 * (u[:, 5] - 0.5)
    return dict(name="config1v_box_hard_volume", world=w, dis=dis, vel=vel, max_rigid=1, steps=2000)


def config4_volume(batch=4096, model="humanoid30.ztk", first=0):
    """the standing humanoid of config 4 under the Volume plugin (what the reference's drivers select): each sole and the
    floor are one rigid pair in volumetric contact - two 6-D wrenches, twelve unknowns"""
    w, dis, vel = _humanoid(batch, "contact_rigid.ztk", B.SOLVER_VOLUME, 0x5EED0004, model, first)
    return dict(name="config4vol_humanoid_rigid_volume", world=w, dis=dis, vel=vel, max_rigid=2, steps=2000)


CONFIGS = {"config1": config1, "config1_volume": config1_volume, "config4_volume": config4_volume, "config1b": config1_rigid, "config2": config2, "config3": config3, "config4": config4, "config4v": config4_vert, "config5": config5, "config5v": config5_vert,
           "config3_26": config3_26, "config4_26": config4_26, "config4_shell": config4_shell}
