"""URDF subset loader: builds a `ModelDescription` from the URDF features the hot path covers.

Conventions follow the reference's dart/utils/urdf/DartLoader.cpp (parameters only; nothing is copied):
  * root link != "world"  -> FreeJoint root (:224-248); a root link named "world" is not a body (:196-222)
  * joint origin -> T_ParentBodyToJoint, T_ChildBodyToJoint = identity (:399-400)
  * inertial origin xyz -> local COM, inertia rotated by the inertial rpy (:527-545)
  * limits/damping per :403-437; `fixed` -> WeldJoint (:482-486), merged into the parent by ModelDescription.merge_welds
  * children are visited depth-first in joint-NAME order (urdfdom keeps joints in a std::map and builds child_links
    from it), which fixes the DOF order of the skeleton
  * collision boxes -> box colliders with the collision origin as the shape's relative transform (:612-616);
    mesh / sphere / capsule colliders (libccd path, not vendored) are dropped
Revolute, continuous, prismatic and fixed joints; joint Coulomb friction and other joint types raise.
"""
import xml.etree.ElementTree as ET

import numpy as np

from .model import BodySpec, BoxSpec, ModelDescription, make_transform


def _floats(s, n=3, default=0.0):
    if s is None:
        return [default] * n
    return [float(x) for x in s.split()]


def _origin(el):
    o = el.find("origin") if el is not None else None
    if o is None:
        return make_transform()
    return make_transform(_floats(o.get("xyz")), _floats(o.get("rpy")))


def load_urdf(path, name=None, weld_joints=()):
    """Parse `path`; joints named in `weld_joints` are frozen at 0 (welded), e.g. the arms of the 20-DOF Atlas."""
    root = ET.parse(path).getroot()
    if name is None:
        name = root.get("name", "model")
    links = {l.get("name"): l for l in root.findall("link")}
    joints = {j.get("name"): j for j in root.findall("joint")}
    child_of = {}
    children = {ln: [] for ln in links}
    for jn in sorted(joints):  # std::map order
        j = joints[jn]
        p, c = j.find("parent").get("link"), j.find("child").get("link")
        child_of[c] = jn
        children[p].append(jn)
    roots = [ln for ln in links if ln not in child_of]
    assert len(roots) == 1, roots
    bodies, boxes = [], []

    def link_inertial(ln):
        inert = links[ln].find("inertial")
        if inert is None:
            return 1.0, (0, 0, 0), (1, 1, 1, 0, 0, 0)  # BodyNode defaults (Inertia.cpp ctor: mass 1, identity moment)
        mass = float(inert.find("mass").get("value"))
        o = inert.find("origin")
        xyz = _floats(o.get("xyz")) if o is not None else [0, 0, 0]
        rpy = _floats(o.get("rpy")) if o is not None else [0, 0, 0]
        i = inert.find("inertia")
        J = np.array([[float(i.get("ixx")), float(i.get("ixy")), float(i.get("ixz"))],
                      [float(i.get("ixy")), float(i.get("iyy")), float(i.get("iyz"))],
                      [float(i.get("ixz")), float(i.get("iyz")), float(i.get("izz"))]])
        R = make_transform((0, 0, 0), rpy)[:3, :3]
        J = R @ J @ R.T
        return mass, tuple(xyz), (J[0, 0], J[1, 1], J[2, 2], J[0, 1], J[0, 2], J[1, 2])

    def add_shapes(ln, body_index):
        for col in links[ln].findall("collision"):
            geom = col.find("geometry")
            box = geom.find("box") if geom is not None else None
            sph = geom.find("sphere") if geom is not None else None
            if box is not None:
                boxes.append(BoxSpec(body_index, _origin(col), tuple(_floats(box.get("size"))), 1.0))
            elif sph is not None:
                r = float(sph.get("radius"))
                boxes.append(BoxSpec(body_index, _origin(col), (r, r, r), 1.0, "sphere"))
            # mesh / capsule / cylinder: outside the analytic box / sphere narrow phase

    def recurse(ln, parent_index):
        for jn in children[ln]:
            j = joints[jn]
            c = j.find("child").get("link")
            jt = j.get("type")
            mass, com, inertia = link_inertial(c)
            kw = {}
            if jt in ("revolute", "continuous", "prismatic") and jn not in weld_joints:
                jtype = "prismatic" if jt == "prismatic" else "revolute"
                lim = j.find("limit")
                if lim is not None and jt != "continuous":
                    kw["pos_lo"] = (float(lim.get("lower", 0.0)),)
                    kw["pos_hi"] = (float(lim.get("upper", 0.0)),)
                if lim is not None:
                    if lim.get("velocity") is not None:
                        kw["vel_lo"] = (-float(lim.get("velocity")),)
                        kw["vel_hi"] = (float(lim.get("velocity")),)
                    if lim.get("effort") is not None:
                        kw["force_lo"] = (-float(lim.get("effort")),)
                        kw["force_hi"] = (float(lim.get("effort")),)
                    lo, hi = float(lim.get("lower", 0.0)), float(lim.get("upper", 0.0))
                    if jt != "continuous" and (lo > 0 or hi < 0):
                        kw["rest"] = ((lo + hi) / 2.0,)  # DartLoader.cpp:414-431
                dyn = j.find("dynamics")
                if dyn is not None:
                    kw["damping"] = (float(dyn.get("damping", 0.0)),)
                    if float(dyn.get("friction", 0.0)) != 0.0:
                        raise ValueError(f"{jn}: joint Coulomb friction is outside the hot-path scope")
                axis = _floats(j.find("axis").get("xyz")) if j.find("axis") is not None else [1, 0, 0]
            elif jt == "fixed" or jn in weld_joints:
                jtype, axis = "weld", [0, 0, 1]
            else:
                raise ValueError(f"{jn}: joint type {jt} outside scope")
            bodies.append(BodySpec(c, parent_index, jtype, jn, axis=tuple(axis), T_pj=_origin(j), mass=mass, com=com,
                                   inertia=inertia, **kw))
            idx = len(bodies) - 1
            add_shapes(c, idx)
            recurse(c, idx)

    rl = roots[0]
    if rl == "world":
        recurse(rl, -1)
    else:
        mass, com, inertia = link_inertial(rl)
        bodies.append(BodySpec(rl, -1, "free", "rootJoint", mass=mass, com=com, inertia=inertia))
        add_shapes(rl, 0)
        recurse(rl, 0)
    return ModelDescription(name, bodies, boxes)


def with_ground(model, ground):
    """One world from two skeletons (the reference loads the robot and the ground URDF into the same World)."""
    nb = len(model.bodies)
    bodies = list(model.bodies)
    boxes = list(model.boxes)
    for b in ground.bodies:
        nbdy = BodySpec(**{**b.__dict__})
        nbdy.parent = b.parent + nb if b.parent >= 0 else -1
        bodies.append(nbdy)
    for bx in ground.boxes:
        boxes.append(BoxSpec(bx.body + nb if bx.body >= 0 else -1, bx.T, bx.size, bx.mu, bx.shape))
    return ModelDescription(model.name + "_ground", bodies, boxes, model.gravity, model.dt, None, max_contacts=8)


