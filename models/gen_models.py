#!/usr/bin/env python3
"""Generate the ZTK model files used by tests and bench (committed under models/).

Provenance:
  * box.ztk, box_small.ztk, floor.ztk, floor_hardsoft.ztk, contactinfo.ztk restate the
    physical parameters of the reference's example models (reference example/model/*.ztk).
  * chain30.ztk is SYNTHETIC (config 2): fixed base + 30 revolute links.
  * humanoid26.ztk keeps the inertial parameters, frames, motors and the two sole shapes of
    the reference's mighty.ztk (reference example/model/mighty.ztk:1690-1738,1766-2399) and
    drops the visual meshes.  humanoid30.ztk is SYNTHETIC (config 3/4): humanoid26 plus a
    4-link waist/neck branch whose inertial parameters are those of left_elbow_rotation.
  * contact_elastic.ztk / contact_rigid.ztk: the single "ground body" entry of configs 3 / 4.
The humanoid files need the reference checkout to regenerate (it only exists in the build
container); all other files are generated from the constants below.

usage: python models/gen_models.py [/root/reference]
"""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def w(name, text):
    with open(os.path.join(HERE, name), "w") as f:
        f.write(text)


def box(name, d, wd, h, mass, inertia, stuff="body"):
    return f"""[roki::chain]
name : {name}

[zeo::shape]
type : box
name : shape
depth : {d}
width : {wd}
height : {h}

[roki::link]
name : link#00
jointtype : float
mass : {mass}
stuff : {stuff}
inertia : {{
 {inertia}, 0, 0
 0, {inertia}, 0
 0, 0, {inertia}
}}
shape : shape
"""


FLOOR = """[roki::chain]
name : floor

[zeo::shape]
type : box
name : shape
center : 0, 0, -0.2
depth : 5.0
width : 5.0
height : 0.4

[roki::link]
name : link#00
jointtype : fixed
mass : 99.9
stuff : ground
inertia : {
 0.999, 0, 0
 0, 0.999, 0
 0, 0, 0.999
}
shape : shape
"""

FLOOR_HARDSOFT = """[roki::chain]
name : floor

[zeo::shape]
type : box
name : shape
center : 0, 0, -0.2
depth : 5.0
width : 2.5
height : 0.4

[zeo::shape]
type : box
name : shape2
center : 0, 0, -0.2
depth : 5.0
width : 2.5
height : 0.4

[roki::link]
name : link#00
jointtype : fixed
mass : 99.9
stuff : ground
COM : { 0, 0, 0 }
inertia : {
 0.999, 0, 0
 0, 0.999, 0
 0, 0, 0.999
}
frame : {
 1, 0, 0, 0
 0, 1, 0, 1.25
 0, 0, 1, 0
}
shape : shape

[roki::link]
name : link#01
jointtype : fixed
mass : 99.9
stuff : soft
COM : { 0, 0, 0 }
inertia : {
 0.999, 0, 0
 0, 0.999, 0
 0, 0, 0.999
}
frame : {
 1, 0, 0, 0
 0, 1, 0, -2.5
 0, 0, 1, 0
}
parent : link#00
shape : shape2
"""


def contact(entries):
    out = []
    for e in entries:
        s = f"[roki::contact]\nbind : {e['bind']}\nstaticfriction : {e['sf']}\nkineticfriction : {e['kf']}\n"
        if "k" in e:
            s += f"compensation : {e['k']}\nrelaxation : {e['l']}\n"
        else:
            s += f"elasticity : {e['e']}\nviscosity : {e['v']}\n"
        out.append(s)
    return "\n".join(out)


CONTACTINFO = [
    dict(bind="ground body", sf=0.5, kf=0.3, k=1000.0, l=0.0001),
    dict(bind="body body", sf=0.5, kf=0.3, k=1000.0, l=0.05),
    dict(bind="wall wall", sf=0.5, kf=0.3, k=500.0, l=0.001),
    dict(bind="wall ground", sf=0.5, kf=0.3, k=500.0, l=0.001),
    dict(bind="ground crawler", sf=10.0, kf=7.0, k=100.0, l=10.0),
    dict(bind="soft body", sf=0.5, kf=0.3, e=100.0, v=1.0),
    dict(bind="soft crawler", sf=10.0, kf=7.0, e=1000.0, v=10.0),
]


def chain30():
    """config 2: fixed base + 30 revolute links, axis alternating z / y, 0.1 m links along x."""
    s = "[roki::chain]\nname : chain30\n\n"
    s += "[roki::link]\nname : base\njointtype : fixed\nmass : 1.0\nstuff : body\n"
    s += "inertia : {\n 1e-3, 0, 0\n 0, 1e-3, 0\n 0, 0, 1e-3\n}\n\n"
    # a frame whose local z is the parent's y flips the rotation axis; applying it on every
    # link alternates the world axis z, y, z, y ... at the zero configuration
    for i in range(30):
        parent = "base" if i == 0 else f"link#{i:02d}"
        px = 0.0 if i == 0 else 0.1
        if i == 0:
            frame = f" 1, 0, 0, {px}\n 0, 1, 0, 0\n 0, 0, 1, 0\n"
        elif i % 2 == 1:
            frame = f" 1, 0, 0, {px}\n 0, 0, 1, 0\n 0, -1, 0, 0\n"
        else:
            frame = f" 1, 0, 0, {px}\n 0, 0, -1, 0\n 0, 1, 0, 0\n"
        s += f"[roki::link]\nname : link#{i+1:02d}\njointtype : revolute\nmass : 1.0\nstuff : body\n"
        s += "COM : { 0.05, 0, 0 }\n"
        s += "inertia : {\n 1e-3, 0, 0\n 0, 8.8e-3, 0\n 0, 0, 8.8e-3\n}\n"
        s += "frame : {\n" + frame + "}\n"
        s += f"parent : {parent}\n\n"
    return s


def arm(name, root_joint):
    """SYNTHETIC test model: a 4-joint arm (yaw, shoulder pitch, telescopic forearm, wrist roll) with a
    box-shaped hand.  root_joint = 'fixed': a static pedestal carries the yaw joint; 'revolute': the yaw
    link is the root itself.  Exercises prismatic joints, both motor types, joint friction and contact
    paths that end at a fixed / 1-DoF root."""
    inertia = "inertia : {\n 1e-2, 0, 0\n 0, 1e-2, 0\n 0, 0, 1e-2\n}\n"
    s = f"[roki::chain]\nname : {name}\n\n"
    s += "[zeo::shape]\ntype : box\nname : hand\ncenter : 0, 0, 0.1\ndepth : 0.08\nwidth : 0.08\nheight : 0.08\n\n"
    s += ("[roki::motor]\nname : dcm\ntype: dc\nmotorconstant : 2.53e-2\nadmittance : 0.045872\nmaxvoltage : 24.0\nminvoltage : -24.0\n"
          "gearratio : 100.0\nrotorinertia : 2.97e-7\ngearinertia : 3.0e-7\n\n")
    s += "[roki::motor]\nname : trqm\ntype: trq\nmax : 20.0\nmin : -20.0\n\n"
    base = " 1, 0, 0, 0\n 0, 1, 0, 0\n 0, 0, 1, 0.4\n"
    if root_joint == "fixed":
        s += "[roki::link]\nname : pedestal\njointtype : fixed\nmass : 5.0\nstuff : body\n" + inertia + "frame : {\n" + base + "}\n\n"
        s += "[roki::link]\nname : yaw\njointtype : revolute\nmass : 1.0\nstuff : body\nCOM : { 0, 0, 0.02 }\n" + inertia
        s += "motor : dcm\nstiffness: 0.0\nviscosity: 0.0\ncoulomb: 1.0\nstaticfriction: 1.2\nparent : pedestal\n\n"
    else:
        s += "[roki::link]\nname : yaw\njointtype : revolute\nmass : 1.0\nstuff : body\nCOM : { 0, 0, 0.02 }\n" + inertia
        s += "motor : dcm\nstiffness: 0.0\nviscosity: 0.0\ncoulomb: 1.0\nstaticfriction: 1.2\nframe : {\n" + base + "}\n\n"
    # shoulder pitch: local z = parent's y
    s += "[roki::link]\nname : shoulder\njointtype : revolute\nmass : 1.5\nstuff : body\nCOM : { 0.15, 0, 0 }\n" + inertia
    s += "motor : trqm\nframe : {\n 1, 0, 0, 0\n 0, 0, 1, 0\n 0, -1, 0, 0\n}\nparent : yaw\n\n"
    # telescopic forearm: local z = parent's x
    s += "[roki::link]\nname : forearm\njointtype : prismatic\nmass : 0.8\nstuff : body\nCOM : { 0, 0, 0.1 }\n" + inertia
    s += "frame : {\n 0, 0, 1, 0.3\n 0, 1, 0, 0\n -1, 0, 0, 0\n}\nparent : shoulder\n\n"
    s += "[roki::link]\nname : wrist\njointtype : revolute\nmass : 0.4\nstuff : body\nCOM : { 0, 0, 0.1 }\n"
    s += "inertia : {\n 1e-3, 0, 0\n 0, 1e-3, 0\n 0, 0, 1e-3\n}\n"
    s += "frame : {\n 1, 0, 0, 0\n 0, 1, 0, 0\n 0, 0, 1, 0.2\n}\nparent : forearm\nshape : hand\n\n"
    return s


def arm_fold():
    """SYNTHETIC test model for SELF-COLLISION (pairs between links of one chain, which registration forms by default:
    reference src/rkfd_sim.c:198, the "self collision" branch of src/rkfd_util.c:163-170): a pedestal with a base block and
    three links in a vertical plane, a box on every link (shorter than the link, so that neighbours do not touch at the
    joints), torque motors on the joints.  Folded (q2 ~ 120 deg, q3 ~ 140 deg) the last link's box comes down on the first
    link's - a rigid 'body body' contact with BOTH sides on the same tree."""
    inertia = "inertia : {\n 1e-3, 0, 0\n 0, 1e-3, 0\n 0, 0, 1e-3\n}\n"
    s = "[roki::chain]\nname : arm_fold\n\n"
    s += "[zeo::shape]\ntype : box\nname : base\ncenter : 0, 0, -0.1\ndepth : 0.1\nwidth : 0.1\nheight : 0.1\n\n"
    s += "[zeo::shape]\ntype : box\nname : beam\ncenter : 0.15, 0, 0\ndepth : 0.18\nwidth : 0.04\nheight : 0.04\n\n"
    s += "[zeo::shape]\ntype : box\nname : finger\ncenter : 0.13, 0, 0\ndepth : 0.2\nwidth : 0.03\nheight : 0.03\n\n"
    s += "[roki::motor]\nname : trqm\ntype: trq\nmax : 20.0\nmin : -20.0\n\n"
    s += "[roki::link]\nname : pedestal\njointtype : fixed\nmass : 5.0\nstuff : body\n" + inertia
    s += "frame : {\n 1, 0, 0, 0\n 0, 1, 0, 0\n 0, 0, 1, 0.5\n}\nshape : base\n\n"
    # local z = -parent's y: a positive joint angle lifts the link's x axis towards +z
    s += "[roki::link]\nname : link1\njointtype : revolute\nmass : 0.5\nstuff : body\nCOM : { 0.15, 0, 0 }\n" + inertia
    s += "motor : trqm\nviscosity: 0.05\nframe : {\n 1, 0, 0, 0\n 0, 0, -1, 0\n 0, 1, 0, 0\n}\nparent : pedestal\nshape : beam\n\n"
    s += "[roki::link]\nname : link2\njointtype : revolute\nmass : 0.5\nstuff : body\nCOM : { 0.15, 0, 0 }\n" + inertia
    s += "motor : trqm\nviscosity: 0.05\nframe : {\n 1, 0, 0, 0.3\n 0, 1, 0, 0\n 0, 0, 1, 0\n}\nparent : link1\nshape : beam\n\n"
    s += "[roki::link]\nname : link3\njointtype : revolute\nmass : 0.3\nstuff : body\nCOM : { 0.1, 0, 0 }\n" + inertia
    s += "motor : trqm\nviscosity: 0.05\nframe : {\n 1, 0, 0, 0.3\n 0, 1, 0, 0\n 0, 0, 1, 0\n}\nparent : link2\nshape : finger\n\n"
    return s


def wall(name, nbrick, thresholds, upright=True, stuff="wall"):
    """SYNTHETIC, after the reference's example/model/wall.ztk: a base block fixed to the world and `nbrick` bricks in a row, each
    hanging on its predecessor by a BREAKABLE FLOAT joint (jointtype: breakablefloat, forcethreshold / torquethreshold); every
    link carries the same brick shape.  upright: the row stands as a column (the base frame turns the links' x axis up, as in
    the reference's file), else it sticks out horizontally (a cantilever: the joints carry bending moments)."""
    inertia = "inertia: {\n 0.0002604166667, 0, 0\n 0, 0.0004166666667, 0\n 0, 0, 0.0002604166667\n}\n"
    s = f"[roki::chain]\nname : {name}\n\n"
    s += "[zeo::shape]\ntype : box\nname : brick\ncenter: ( 0.05 0 0 )\ndepth: 0.099\nwidth: 0.049\nheight: 0.099\n\n"
    base = " 0, 0, -1, 0\n 0, 1, 0, 0.55\n 1, 0, 0, 0\n" if upright else " 1, 0, 0, 0\n 0, 1, 0, 0.55\n 0, 0, 1, 0.5\n"
    s += f"[roki::link]\nname: base\njointtype: fixed\nmass: 0.25\nCOM: ( 0.05 0 0 )\n{inertia}stuff: {stuff}\nframe: {{\n{base}}}\nshape: brick\n\n"
    for k in range(nbrick):
        f, t = thresholds[k]
        par = "base" if k == 0 else f"brick{k}"
        s += (f"[roki::link]\nname: brick{k+1}\njointtype: breakablefloat\nforcethreshold: {f}\ntorquethreshold: {t}\nmass: 0.25\n"
              f"COM: ( 0.05 0 0 )\n{inertia}stuff: {stuff}\nframe: {{\n 1, 0, 0, 0.1\n 0, 1, 0, 0\n 0, 0, 1, 0\n}}\nshape: brick\nparent: {par}\n\n")
    return s


def humanoid(ref_root):
    src = os.path.join(ref_root, "example", "model", "mighty.ztk")
    text = open(src).read()
    # split into tagged sections
    parts = re.split(r"(?m)^(?=\[)", text)
    keep = []
    for p in parts:
        tag = p.split("]", 1)[0].strip("[") if p.startswith("[") else ""
        if tag == "roki::chain":
            keep.append("[roki::chain]\nname : humanoid26\n\n")
        elif tag == "zeo::shape":
            m = re.search(r"(?m)^name\s*:\s*(\S+)", p)
            if m and m.group(1) in ("left_sole", "right_sole"):
                body = "\n".join(l for l in p.splitlines() if not l.strip().startswith("optic") and not l.strip().startswith("%"))
                keep.append(body.rstrip() + "\n\n")
        elif tag == "roki::motor":
            keep.append("\n".join(l for l in p.splitlines() if not l.strip().startswith("%")).rstrip() + "\n\n")
        elif tag == "roki::link":
            lines = []
            for l in p.splitlines():
                t = l.strip()
                if t.startswith("%"):
                    continue
                if t.startswith("shape"):
                    nm = t.split(":", 1)[1].strip()
                    if nm not in ("left_sole", "right_sole"):
                        continue
                lines.append(l)
            keep.append("\n".join(lines).rstrip() + "\n\n")
        elif tag == "roki::chain::init":
            keep.append("\n".join(l for l in p.splitlines() if not l.strip().startswith("%")).rstrip() + "\n")
    h26 = "".join(keep)
    header = ("% humanoid26: inertial parameters, frames, motors and sole shapes of the reference's\n"
              "% example/model/mighty.ztk (visual meshes dropped).  Generated by models/gen_models.py.\n")
    w("humanoid26.ztk", header + h26)

    # humanoid30: + waist_yaw, waist_pitch, neck_yaw, neck_pitch (serial branch off body)
    elbow = re.search(r"(?s)\[roki::link\]\nname: left_elbow_rotation.*?(?=\[roki::link\])", text).group(0)
    mass = re.search(r"mass:\s*(\S+)", elbow).group(1)
    com = re.search(r"COM:\s*(\{.*?\})", elbow).group(1)
    inertia = re.search(r"(?s)inertia:\s*(\{.*?\})", elbow).group(1)
    extra = []
    frames = {
        "waist_yaw":   (" 1, 0, 0, 0\n 0, 1, 0, 0\n 0, 0, 1, 0.05\n", "body"),
        "waist_pitch": (" 1, 0, 0, 0\n 0, 0, 1, 0\n 0, -1, 0, 0.03\n", "waist_yaw"),
        "neck_yaw":    (" 1, 0, 0, 0\n 0, 0, -1, 0.05\n 0, 1, 0, 0\n", "waist_pitch"),
        "neck_pitch":  (" 1, 0, 0, 0\n 0, 0, 1, 0\n 0, -1, 0, 0.03\n", "neck_yaw"),
    }
    for nm, (fr, par) in frames.items():
        extra.append(
            f"[roki::link]\nname: {nm}\njointtype: revolute\nmax: 90\nmin:-90\nstiffness: 0.0\nviscosity: 0.0\n"
            f"coulomb: 1.0\nstaticfriction: 1.2\nmotor: RE-max17\nmass: {mass}\nstuff: body\nCOM: {com}\n"
            f"inertia: {inertia}\nframe: {{\n{fr}}}\nparent: {par}\n\n")
    h30 = h26.replace("name : humanoid26", "name : humanoid30")
    idx = h30.index("[roki::chain::init]")
    h30 = h30[:idx] + "".join(extra) + h30[idx:]
    header30 = ("% humanoid30 (SYNTHETIC, SURVEY.md 8d config 3): humanoid26 plus a 4-link waist/neck branch\n"
                "% (inertial parameters of left_elbow_rotation).  Generated by models/gen_models.py.\n")
    w("humanoid30.ztk", header30 + h30)


def humanoid_shell():
    """humanoid30_shell.ztk (SYNTHETIC): humanoid30 with a tessellated sphere (div 16: 114 vertices) on six links, so that
    the world holds ~760 candidate contact vertices like the reference's mighty.ztk with its body meshes (749): the
    multi-chunk collision sweep with a humanoid.  Generated from the committed humanoid30.ztk (no reference needed)."""
    text = open(os.path.join(HERE, "humanoid30.ztk")).read()
    links = ["body", "left_knee_flexion", "right_knee_flexion", "left_elbow_flexion", "right_elbow_flexion", "neck_pitch"]
    shapes = "".join(f"[zeo::shape]\nname: shell_{l}\ntype: sphere\ncenter: 0, 0, 0\nradius: 0.03\ndiv: 16\n\n" for l in links)
    text = text.replace("[zeo::shape]\nname: left_sole", shapes + "[zeo::shape]\nname: left_sole", 1)
    for l in links:
        text = re.sub(r"(\[roki::link\]\nname: %s\n)" % l, r"\1shape: shell_%s\n" % l, text, count=1)
    text = text.replace("name : humanoid30", "name : humanoid30_shell", 1)
    text = text.replace("% humanoid30 (SYNTHETIC", "% humanoid30_shell: humanoid30 + six tessellated spheres (models/gen_models.py: humanoid_shell)\n% humanoid30 (SYNTHETIC", 1)
    assert text.count("shape: shell_") == len(links)
    w("humanoid30_shell.ztk", text)


def lfoot():
    """lfoot.ztk (SYNTHETIC): one free body with a CONVEX foot plate (box, below) and a NON-CONVEX bracket above it (an L-shaped
    prism as a polyhedron: 12 vertices, 20 triangles, one reflex edge) - what the reference's mighty.ztk is to the Volume plugin
    in the small: it stands on a convex shape the plugin can clip, while the pairs of the shape that is not convex are guarded
    (status 4 when the bracket itself touches something)."""
    # L cross-section in the x-z plane (counter-clockwise), extruded along y
    prof = [(-0.05, 0.01), (0.05, 0.01), (0.05, 0.04), (-0.02, 0.04), (-0.02, 0.12), (-0.05, 0.12)]
    ys = (-0.03, 0.03)
    verts = [(x, y, z) for y in ys for (x, z) in prof]
    n = len(prof)
    faces = []
    for i in range(n):                                   # side walls (outward for a counter-clockwise profile seen from -y)
        j = (i + 1) % n
        faces += [(i, j, n + j), (i, n + j, n + i)]
    tri = [(0, 1, 2), (0, 2, 3), (0, 3, 4), (0, 4, 5)]     # the L as a fan from its inner corner's opposite vertex
    faces += [(a, c, b) for a, b, c in tri]              # cap at y = -0.03 (normal -y)
    faces += [(n + a, n + b, n + c) for a, b, c in tri]  # cap at y = +0.03 (normal +y)
    # orientation check: every face normal points away from a point inside the thick leg
    import numpy as np
    V = np.array(verts); inside = np.array([-0.035, 0.0, 0.025])
    out = []
    for a, b, c in faces:
        nrm = np.cross(V[b] - V[a], V[c] - V[a])
        if np.dot(nrm, V[a] - inside) < 0:
            a, b, c = a, c, b
        out.append((a, b, c))
    vtxt = "".join("vert: %d { %.5f, %.5f, %.5f }\n" % (i, *v) for i, v in enumerate(verts))
    ftxt = "".join("face: %d %d %d\n" % f for f in out)
    return f"""% lfoot (SYNTHETIC, models/gen_models.py: lfoot): a convex foot plate under a non-convex L bracket
[roki::chain]
name : lfoot

[zeo::shape]
type : box
name : plate
center : 0, 0, 0.005
depth : 0.12
width : 0.08
height : 0.01

[zeo::shape]
name: bracket
type: polyhedron
{vtxt}{ftxt}
[roki::link]
name : link#00
jointtype : float
mass : 0.4
stuff : body
COM : 0, 0, 0.03
inertia : {{
 6e-4, 0, 0
 0, 6e-4, 0
 0, 0, 6e-4
}}
shape : plate
shape : bracket
"""


def main():
    w("box.ztk", box("box", 0.1, 0.1, 0.1, 0.5, "8.33e-4"))
    w("box_small.ztk", box("box_small", 0.05, 0.10, 0.05, 0.125, "5.208333333e-05"))
    w("floor.ztk", FLOOR)
    w("floor_hardsoft.ztk", FLOOR_HARDSOFT)
    w("contactinfo.ztk", contact(CONTACTINFO))
    w("contact_elastic.ztk", contact([dict(bind="ground body", sf=0.5, kf=0.3, e=1000.0, v=10.0)]))
    w("contact_rigid.ztk", contact([dict(bind="ground body", sf=0.5, kf=0.3, k=1000.0, l=0.0001)]))
    w("chain30.ztk", chain30())
    w("arm_fixedroot.ztk", arm("arm_fixedroot", "fixed"))
    w("arm_revroot.ztk", arm("arm_revroot", "revolute"))
    w("arm_fold.ztk", arm_fold())
    w("wall.ztk", wall("wall", 3, [(200.0, 200.0), (10.0, 10.0), (10.0, 10.0)]))          # the reference's wall.ztk: same structure and thresholds
    w("wall_cantilever.ztk", wall("wall_cantilever", 2, [(100.0, 0.4), (100.0, 100.0)], upright=False))
    w("lfoot.ztk", lfoot())
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    if os.path.isdir(ref):
        humanoid(ref)
    else:
        print("reference checkout not found: humanoid26/30.ztk left untouched")
    humanoid_shell()


if __name__ == "__main__":
    main()
