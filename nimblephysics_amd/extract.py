"""Model extraction from a LIVE `nimblephysics.simulation.World`: walk the world through the
reference's own Python bindings and build the `ModelDescription` that `nimblephysics_amd.World` uploads - so that a user who
already holds a `nimble.simulation.World` (loaded by the reference's SkelParser / DartLoader) can switch the timestep over
without describing the model again:

    import nimblephysics as nimble, nimblephysics_amd as na
    ref_world = nimble.loadWorld("half_cheetah.skel")
    world = na.World(na.model_from_nimble_world(ref_world))

Only bound methods are used (python/_nimblephysics/{simulation_and_neural/World.cpp, dynamics/{Skeleton,BodyNode,Joint,
RevoluteJoint,PrismaticJoint,ShapeNode,Shape}.cpp}):
  World      getNumSkeletons, getSkeleton, getTimeStep, getGravity, getActionSpace, getContactClippingDepth,
             getFallbackConstraintForceMixingConstant, clone, tuneMass, getMasses
  Skeleton   getNumBodyNodes, getBodyNode
  BodyNode   getName, getParentBodyNode, getParentJoint, getMass, getLocalCOM, getFrictionCoeff, getRestitutionCoeff, getNumShapeNodes, getShapeNode
  Joint      getType, getName, getNumDofs, getTransformFromParentBodyNode, getTransformFromChildBodyNode, getAxis (revolute /
             prismatic), getAxisOrder / getFlipAxisMap (Euler), getAxis1 / getAxis2 (universal), getTranslationalAxis1 / 2 (planar,
             translational-2D), getDampingCoefficient, getSpringStiffness, getRestPosition, get{Position,Velocity,ControlForce}{Lower,Upper}Limit
  ShapeNode  getShape, getRelativeTranslation, getRelativeRotation;  Shape getType, getSize (BoxShape), getRadius (SphereShape)
The bindings expose no getter for a body's moment of inertia (`getMomentOfInertia` takes six C++ reference arguments), so it
is read the way the reference's own Python users read it: on a CLONE of the world every body is registered with
`tuneMass(body, WrtMassBodyNodeEntryType.INERTIA_FULL, ...)` and `getMasses()` returns [mass, com (3), Ixx, Iyy, Izz, Ixy, Ixz,
Iyz] per body (dart/neural/WithRespectToMass.cpp:25-140); the caller's world is left untouched.
The nimblephysics package cannot be built in this repo's environment; tests/test_extract.py drives this walk with a duck-typed
stand-in that exposes exactly the methods above."""
from __future__ import annotations

from typing import Optional

import numpy as np

from .model import BodySpec, BoxSpec, ModelDescription

_JOINT_TYPES = {"RevoluteJoint": "revolute", "PrismaticJoint": "prismatic", "FreeJoint": "free", "WeldJoint": "weld", "BallJoint": "ball", "ScrewJoint": "screw"}
_COMPOUND_TYPES = ("EulerJoint", "UniversalJoint", "TranslationalJoint", "TranslationalJoint2D", "PlanarJoint")
_UNIT = {"x": (1.0, 0.0, 0.0), "y": (0.0, 1.0, 0.0), "z": (0.0, 0.0, 1.0)}


def _mat4(T) -> np.ndarray:
    """Eigen::Isometry3s as bound by eigen_geometry_pybind.cpp (.matrix()) or a plain 4 x 4 array."""
    M = T.matrix() if hasattr(T, "matrix") and callable(T.matrix) else T
    return np.asarray(M, dtype=np.float64).reshape(4, 4)


def _limit(v, lo: bool):
    v = float(v)
    return v if np.isfinite(v) else (-np.inf if lo else np.inf)


def model_from_nimble_world(world, name: str = "extracted", max_contacts: Optional[int] = None, inertia_entry_type=None) -> ModelDescription:
    """`world`: a live nimblephysics.simulation.World (or anything exposing the methods listed in the module docstring).
    inertia_entry_type: nimblephysics.neural.WrtMassBodyNodeEntryType.INERTIA_FULL (looked up when nimblephysics is importable)."""
    if inertia_entry_type is None:
        try:
            import nimblephysics as _nimble       # only when the reference package is installed
            inertia_entry_type = _nimble.neural.WrtMassBodyNodeEntryType.INERTIA_FULL
        except Exception:
            inertia_entry_type = "INERTIA_FULL"
    # moments of inertia through the mass vector of a clone (see the docstring)
    probe = world.clone()
    order = []
    for si in range(probe.getNumSkeletons()):
        sk = probe.getSkeleton(si)
        for bi in range(sk.getNumBodyNodes()):
            b = sk.getBodyNode(bi)
            probe.tuneMass(b, inertia_entry_type, np.full(10, np.inf), np.full(10, -np.inf))
            order.append((si, bi))
    masses = np.asarray(probe.getMasses(), dtype=np.float64).reshape(len(order), 10)
    inertia_of = {key: masses[i] for i, key in enumerate(order)}

    bodies, boxes = [], []
    for si in range(world.getNumSkeletons()):
        sk = world.getSkeleton(si)
        # Skeleton::isEnabledSelfCollisionCheck / isEnabledAdjacentBodyCheck (python/_nimblephysics/dynamics/Skeleton.cpp:505-533)
        self_col = bool(sk.isEnabledSelfCollisionCheck()) if hasattr(sk, "isEnabledSelfCollisionCheck") else False
        adj_col = bool(sk.isEnabledAdjacentBodyCheck()) if hasattr(sk, "isEnabledAdjacentBodyCheck") else False
        index = {}
        for bi in range(sk.getNumBodyNodes()):                 # skeleton order = parents before children = DOF order
            b = sk.getBodyNode(bi)
            j = b.getParentJoint()
            jt = j.getType()
            if jt not in _JOINT_TYPES and jt not in _COMPOUND_TYPES:
                raise ValueError(f"{b.getName()}: joint type {jt} outside the hot-path scope (revolute, prismatic, free, ball, weld, "
                                 "Euler, universal, translational, translational-2D, planar)")
            parent = b.getParentBodyNode()
            pidx = -1 if parent is None else index[parent.getName()]
            nd = int(j.getNumDofs())
            kw = {}
            axis = (0.0, 0.0, 1.0)
            vec3 = lambda x: tuple(float(c) for c in np.asarray(x).reshape(3))

            def per_dof():
                return dict(damping=tuple(float(j.getDampingCoefficient(k)) for k in range(nd)),
                            spring=tuple(float(j.getSpringStiffness(k)) for k in range(nd)),
                            rest=tuple(float(j.getRestPosition(k)) for k in range(nd)),
                            pos_lo=tuple(_limit(j.getPositionLowerLimit(k), True) for k in range(nd)),
                            pos_hi=tuple(_limit(j.getPositionUpperLimit(k), False) for k in range(nd)),
                            vel_lo=tuple(_limit(j.getVelocityLowerLimit(k), True) for k in range(nd)),
                            vel_hi=tuple(_limit(j.getVelocityUpperLimit(k), False) for k in range(nd)),
                            force_lo=tuple(_limit(j.getControlForceLowerLimit(k), True) for k in range(nd)),
                            force_hi=tuple(_limit(j.getControlForceUpperLimit(k), False) for k in range(nd)),
                            # Joint::isPositionLimitEnforced (python/_nimblephysics/dynamics/Joint.cpp:202): joint-limit LCP rows
                            limit_enforced=bool(j.isPositionLimitEnforced()) if hasattr(j, "isPositionLimitEnforced") else False)
            if jt in _COMPOUND_TYPES:
                # expanded into 1-DOF chains by ModelDescription (model.py); only the axes differ per class
                kw = per_dof()
                if jt == "EulerJoint":
                    order = j.getAxisOrder()
                    order = getattr(order, "name", str(order)).split(".")[-1].lower()      # AxisOrder.XYZ -> "xyz"
                    flip = np.asarray(j.getFlipAxisMap(), dtype=np.float64).reshape(3)
                    jtype = "euler_" + order
                    kw["axes"] = [tuple(float(flip[k]) * e for e in _UNIT[c]) for k, c in enumerate(order)]
                elif jt == "UniversalJoint":
                    jtype, kw["axes"] = "universal", [vec3(j.getAxis1()), vec3(j.getAxis2())]
                elif jt == "TranslationalJoint":
                    jtype = "translational"
                else:
                    jtype = "planar" if jt == "PlanarJoint" else "translational2d"
                    kw["axes"] = [vec3(j.getTranslationalAxis1()), vec3(j.getTranslationalAxis2())]
            else:
                jtype = _JOINT_TYPES[jt]
                if jtype in ("revolute", "prismatic", "screw"):
                    axis = vec3(j.getAxis())
                    kw = per_dof()
                    if jtype == "screw":
                        kw["pitch"] = float(j.getPitch())
                elif jtype == "free":
                    kw = {k_: v for k_, v in per_dof().items() if k_ in ("damping", "spring", "rest")}
                elif jtype == "ball":
                    kw = per_dof()
            m = inertia_of[(si, bi)]
            bodies.append(BodySpec(b.getName(), pidx, jtype, j.getName(), axis=axis,
                                   T_pj=_mat4(j.getTransformFromParentBodyNode()), T_cj=_mat4(j.getTransformFromChildBodyNode()),
                                   mass=float(b.getMass()), com=tuple(float(x) for x in np.asarray(b.getLocalCOM()).reshape(3)),
                                   inertia=tuple(float(x) for x in m[4:10]), skeleton=si, self_collision=self_col, adjacent_body_check=adj_col, **kw))
            gidx = len(bodies) - 1
            index[b.getName()] = gidx
            for k in range(int(b.getNumShapeNodes())):
                sn = b.getShapeNode(k)
                # the loaders of the reference create separate visual-only and collision shape nodes for the same geometry
                # (SkelParser.cpp:612-640 createShapeNodeWith<VisualAspect> / <CollisionAspect, DynamicsAspect>, DartLoader.cpp): only
                # nodes with a CollisionAspect are collision objects (ShapeFrame.cpp: DARTPY_DEFINE_SPECIALIZED_ASPECT(CollisionAspect))
                if not sn.hasCollisionAspect():
                    continue
                shp = sn.getShape()
                T = np.eye(4)
                T[:3, :3] = np.asarray(sn.getRelativeRotation(), dtype=np.float64).reshape(3, 3)
                T[:3, 3] = np.asarray(sn.getRelativeTranslation(), dtype=np.float64).reshape(3)
                mu = float(b.getFrictionCoeff())
                e = float(b.getRestitutionCoeff())
                if shp.getType() == "BoxShape":
                    boxes.append(BoxSpec(gidx, T, tuple(float(x) for x in np.asarray(shp.getSize()).reshape(3)), mu, "box", e))
                elif shp.getType() == "SphereShape":
                    r = float(shp.getRadius())
                    boxes.append(BoxSpec(gidx, T, (r, r, r), mu, "sphere", e))
                elif shp.getType() == "CapsuleShape":
                    boxes.append(BoxSpec(gidx, T, (float(shp.getRadius()), float(shp.getHeight()), 0.0), mu, "capsule", e))
                # meshes, cylinders, ...: outside the analytic narrow phases (dropped, like in the loaders)
    g = tuple(float(x) for x in np.asarray(world.getGravity()).reshape(3))
    md = ModelDescription(name, bodies, boxes, g, float(world.getTimeStep()), None, max_contacts=(max_contacts or 0) if boxes else 0,
                          contact_clipping_depth=float(world.getContactClippingDepth()),
                          fallback_cfm=float(world.getFallbackConstraintForceMixingConstant()),
                          penetration_correction=bool(world.getPenetrationCorrectionEnabled()))
    if boxes and max_contacts is None:          # (not said: by what the world's collider pairs can hold)
        md.max_contacts = md.suggest_max_contacts()
    aspace = [int(a) for a in world.getActionSpace()]
    if aspace != list(range(md.num_dofs)):
        md.set_action_space(aspace)
    return md
