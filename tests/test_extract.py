"""Model extraction from a live `nimblephysics.simulation.World` (nimblephysics_amd/extract.py).  The reference package cannot be
built here, so the walk is driven by a duck-typed stand-in that exposes EXACTLY the bound methods the walk uses (names and
call shapes taken from python/_nimblephysics/**): a world built from one of our own descriptions must come back unchanged."""
import copy

import numpy as np
import pytest

import nimblephysics_amd as na
from nimblephysics_amd.extract import model_from_nimble_world


class _Iso:                      # Eigen::Isometry3s as bound by eigen_geometry_pybind.cpp
    def __init__(self, T): self._T = np.array(T, dtype=np.float64)
    def matrix(self): return self._T.copy()


class _Shape:
    def __init__(self, bx): self._bx = bx
    def getType(self): return {"sphere": "SphereShape", "capsule": "CapsuleShape"}.get(self._bx.shape, "BoxShape")
    def getHeight(self): return float(self._bx.size[1])
    def getSize(self): return np.array(self._bx.size, dtype=np.float64)
    def getRadius(self): return float(self._bx.size[0])


class _ShapeNode:
    """The reference's loaders give a body one visual-only ShapeNode AND one collision ShapeNode per geometry (SkelParser.cpp:612-640,
    DartLoader.cpp createShapeNodeWith<VisualAspect> / <CollisionAspect, DynamicsAspect>): the stand-in does the same."""
    def __init__(self, bx, collision=True): self._bx, self._collision = bx, collision
    def hasCollisionAspect(self): return self._collision
    def hasVisualAspect(self): return not self._collision
    def getShape(self): return _Shape(self._bx)
    def getRelativeTranslation(self): return np.array(self._bx.T)[:3, 3].copy()
    def getRelativeRotation(self): return np.array(self._bx.T)[:3, :3].copy()


class _Joint:
    TYPES = {"revolute": "RevoluteJoint", "prismatic": "PrismaticJoint", "free": "FreeJoint", "weld": "WeldJoint",
             "universal": "UniversalJoint", "translational": "TranslationalJoint", "translational2d": "TranslationalJoint2D",
             "planar": "PlanarJoint", "ball": "BallJoint"}

    def __init__(self, b): self._b = b
    def getType(self): return "EulerJoint" if self._b.joint_type.startswith("euler_") else self.TYPES[self._b.joint_type]
    def getName(self): return self._b.joint_name
    def isPositionLimitEnforced(self): return bool(self._b.limit_enforced)
    def getNumDofs(self):
        from nimblephysics_amd.model import COMPOUND_JOINTS
        return COMPOUND_JOINTS.get(self._b.joint_type, {"free": 6, "weld": 0, "ball": 3}.get(self._b.joint_type, 1))
    # the class-specific getters of the compound joints (python/_nimblephysics/dynamics/{Euler,Universal,Planar,TranslationalJoint2D}Joint.cpp)
    def getAxisOrder(self):
        import enum
        return enum.Enum("AxisOrder", "XYZ XZY ZYX ZXY")[self._b.joint_type[6:].upper()]
    def getFlipAxisMap(self): return np.array(self._flip)
    _flip = (1.0, 1.0, 1.0)
    def getAxis1(self): return np.array(self._b.axes[0], dtype=np.float64)
    def getAxis2(self): return np.array(self._b.axes[1], dtype=np.float64)
    def getTranslationalAxis1(self): return np.array(self._b.axes[0], dtype=np.float64)
    def getTranslationalAxis2(self): return np.array(self._b.axes[1], dtype=np.float64)
    def getTransformFromParentBodyNode(self): return _Iso(self._b.T_pj)
    def getTransformFromChildBodyNode(self): return _Iso(self._b.T_cj)
    def getAxis(self): return np.array(self._b.axis, dtype=np.float64)
    def _get(self, field, k, default): v = getattr(self._b, field); return float(v[k]) if len(v) > k else default
    def getDampingCoefficient(self, k): return self._get("damping", k, 0.0)
    def getSpringStiffness(self, k): return self._get("spring", k, 0.0)
    def getRestPosition(self, k): return self._get("rest", k, 0.0)
    def getPositionLowerLimit(self, k): return self._get("pos_lo", k, -np.inf)
    def getPositionUpperLimit(self, k): return self._get("pos_hi", k, np.inf)
    def getVelocityLowerLimit(self, k): return self._get("vel_lo", k, -np.inf)
    def getVelocityUpperLimit(self, k): return self._get("vel_hi", k, np.inf)
    def getControlForceLowerLimit(self, k): return self._get("force_lo", k, -np.inf)
    def getControlForceUpperLimit(self, k): return self._get("force_hi", k, np.inf)


class _Body:
    def __init__(self, md, i): self._md, self._i = md, i
    @property
    def _b(self): return self._md.bodies[self._i]
    def getName(self): return self._b.name
    def getParentJoint(self): return _Joint(self._b)
    def getParentBodyNode(self): return None if self._b.parent < 0 else _Body(self._md, self._b.parent)
    def getMass(self): return float(self._b.mass)
    def getLocalCOM(self): return np.array(self._b.com, dtype=np.float64)
    def getFrictionCoeff(self):
        mus = [bx.mu for bx in self._md.boxes if bx.body == self._i]
        return float(mus[0]) if mus else 1.0
    def getRestitutionCoeff(self):
        es = [bx.restitution for bx in self._md.boxes if bx.body == self._i]
        return float(es[0]) if es else 0.0
    def _shape_nodes(self):        # visual-only nodes first, like the loaders create them
        mine = [bx for bx in self._md.boxes if bx.body == self._i]
        return [_ShapeNode(bx, collision=False) for bx in mine] + [_ShapeNode(bx) for bx in mine]
    def getNumShapeNodes(self): return len(self._shape_nodes())
    def getShapeNode(self, k): return self._shape_nodes()[k]


class _Skeleton:
    def __init__(self, md, idxs): self._md, self._idxs = md, idxs
    def getNumBodyNodes(self): return len(self._idxs)
    def getBodyNode(self, i): return _Body(self._md, self._idxs[i])
    def isEnabledSelfCollisionCheck(self): return bool(self._md.bodies[self._idxs[0]].self_collision)
    def isEnabledAdjacentBodyCheck(self): return bool(self._md.bodies[self._idxs[0]].adjacent_body_check)


class StandInWorld:
    """The slice of nimblephysics.simulation.World the extraction walks, backed by one of our descriptions: every root body
    starts a skeleton."""

    def __init__(self, md):
        self._md = md
        root_of = {}
        for i, b in enumerate(md.bodies):
            root_of[i] = i if b.parent < 0 else root_of[b.parent]
        roots = sorted(set(root_of.values()))
        self._skels = [[i for i in range(len(md.bodies)) if root_of[i] == r] for r in roots]
        self._tuned = []

    def clone(self): return StandInWorld(copy.deepcopy(self._md))
    def getNumSkeletons(self): return len(self._skels)
    def getSkeleton(self, i): return _Skeleton(self._md, self._skels[i])
    def getTimeStep(self): return self._md.dt
    def getGravity(self): return np.array(self._md.gravity)
    def getActionSpace(self): return list(self._md.action_map)
    def getContactClippingDepth(self): return self._md.contact_clipping_depth
    def getFallbackConstraintForceMixingConstant(self): return self._md.fallback_cfm
    def getPenetrationCorrectionEnabled(self): return self._md.penetration_correction
    def tuneMass(self, body, entry_type, upper, lower):
        assert entry_type == "INERTIA_FULL" and len(upper) == 10 and len(lower) == 10
        self._tuned.append(body._i)

    def getMasses(self):           # WithRespectToMass::get for INERTIA_FULL entries: mass, com, Ixx Iyy Izz Ixy Ixz Iyz
        out = []
        for i in self._tuned:
            b = self._md.bodies[i]
            out += [b.mass, *b.com, *b.inertia]
        return np.array(out, dtype=np.float64)


def _same(a, b):
    fa, fb = a.flat(), b.flat()
    assert fa.keys() == fb.keys()
    for k in fa:
        x, y = fa[k], fb[k]
        if k == "body_skeleton":     # the ids are arbitrary, the partition of the bodies into skeletons is what counts
            assert [[i == j for j in x] for i in x] == [[i == j for j in y] for i in y]
            continue
        if isinstance(x, (str, type(None))) or isinstance(y, (str, type(None))):
            assert x == y, k
        else:
            assert np.allclose(np.asarray(x, dtype=float), np.asarray(y, dtype=float), equal_nan=True), k


@pytest.mark.parametrize("make", [lambda: na.cartpole(), lambda: na.single_pendulum(), lambda: na.box_stack(),
                                  lambda: na.atlas("atlas20", ground=True), lambda: na.atlas("atlas33"),
                                  lambda: __import__("test_ball_joint").ball_model(0, True)])
def test_extraction_round_trip(make):
    md = make()
    got = model_from_nimble_world(StandInWorld(md), name=md.name, max_contacts=md.max_contacts)
    assert [b.name for b in got.bodies] == [b.name for b in md.bodies]
    assert [b.joint_type for b in got.bodies] == [b.joint_type for b in md.bodies]
    _same(got, md)
    if md.has_welds():
        _same(got.merge_welds(), md.merge_welds())


def test_extraction_keeps_spheres_friction_action_space_and_leaves_the_world_untouched():
    from nimblephysics_amd.model import BodySpec, BoxSpec, ModelDescription, SphereSpec, make_transform
    I = 0.4 * 0.1 * 0.1
    bodies = [BodySpec("slab", -1, "weld", "fix", T_pj=make_transform((0, -0.5, 0))),
              BodySpec("ball0", -1, "free", "ball0_joint", mass=1.0, inertia=(I, I, I, 0, 0, 0)),
              BodySpec("arm", 1, "revolute", "arm_joint", axis=(0, 0, 1), T_pj=make_transform((0.15, 0, 0)), T_cj=make_transform((-0.15, 0, 0)),
                       mass=0.5, inertia=(0.002, 0.002, 0.002, 0, 0, 0), damping=(0.2,), pos_lo=(-1.0,), pos_hi=(2.0,)),
              BodySpec("ball1", -1, "free", "ball1_joint", mass=2.0, inertia=(2 * I, 2 * I, 2 * I, 0, 0, 0))]
    boxes = [BoxSpec(0, np.eye(4), (4.0, 1.0, 4.0), 0.3, "box", 0.9), SphereSpec(1, np.eye(4), 0.1, 0.8), SphereSpec(2, make_transform((0.05, 0, 0)), 0.1, 0.8),
             SphereSpec(3, np.eye(4), 0.1, 0.8)]
    md = ModelDescription("balls", bodies, boxes, max_contacts=8)
    md.set_action_space([5, 6, 12])
    w = StandInWorld(md)
    got = model_from_nimble_world(w, name=md.name, max_contacts=md.max_contacts)
    assert w._tuned == []                                   # the mass vector was registered on a clone only
    assert [bx.shape for bx in got.boxes] == [bx.shape for bx in md.boxes] and got.boxes[0].mu == 0.3
    assert list(got.action_map) == [5, 6, 12]
    _same(got, md)


def test_extraction_keeps_the_position_limit_enforcement_flag():
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from util import limited_arm
    md = limited_arm()
    md.bodies[2].limit_enforced = False
    got = model_from_nimble_world(StandInWorld(md), name=md.name, max_contacts=8)
    assert [b.limit_enforced for b in got.bodies] == [True, True, False, True, True]
    assert got.flat()["dof_limit_enforced"].tolist() == [1, 1, 0, 1, 1]
    _same(got, md)


def test_extraction_keeps_the_self_collision_flags_of_the_skeletons():
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from util import folding_arm
    for adjacent in (False, True):
        md = folding_arm(True, "sphere", adjacent=adjacent)
        got = model_from_nimble_world(StandInWorld(md), name=md.name, max_contacts=8)
        assert all(b.self_collision and b.adjacent_body_check == adjacent for b in got.bodies)
        assert got.flat()["body_self_collision"].tolist() == [3 if adjacent else 1] * 3
        _same(got, md)
    assert model_from_nimble_world(StandInWorld(folding_arm(False)), name="x", max_contacts=8).flat()["body_self_collision"].tolist() == [0, 0, 0]


def test_extraction_keeps_capsules():
    """CapsuleShape (getRadius / getHeight, python/_nimblephysics/dynamics/Shape.cpp:769-782) -> capsule colliders."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from util import capsule_world
    md = capsule_world(order="free_first", kinds=("capsule", "sphere"))
    md.bodies.append(na.BodySpec("fixed", -1, "weld", "fix"))          # a live World has no body-less colliders: the fixed capsule on a welded body
    md.boxes[-1].body = len(md.bodies) - 1
    got = model_from_nimble_world(StandInWorld(md), name=md.name, max_contacts=md.max_contacts)
    assert [bx.shape for bx in got.boxes] == ["capsule", "sphere", "capsule"]
    assert [tuple(bx.size) for bx in got.boxes] == [tuple(bx.size) for bx in md.boxes]
    _same(got, md)


def test_unsupported_joint_types_raise():
    md = na.cartpole()
    w = StandInWorld(md)
    _Joint.TYPES = dict(_Joint.TYPES, revolute="CustomJoint1")
    try:
        with pytest.raises(ValueError):
            model_from_nimble_world(w)
    finally:
        _Joint.TYPES = dict(_Joint.TYPES, revolute="RevoluteJoint")


def test_compound_joints_are_extracted_through_their_class_getters():
    """Euler (axis order + flip map), universal, planar, translational-2D and translational joints of a live world come back as
    the same 1-DOF chains the SKEL loader builds for them."""
    import os
    import types
    from test_compound_joints import _skel_bodies
    raw, md = _skel_bodies()
    shell = types.SimpleNamespace(bodies=raw, boxes=[], dt=md.dt, gravity=md.gravity, action_map=list(range(md.num_dofs)),
                                  contact_clipping_depth=md.contact_clipping_depth, fallback_cfm=md.fallback_cfm,
                                  penetration_correction=False)
    got = model_from_nimble_world(StandInWorld(shell), name=md.name, max_contacts=0)
    assert [b.joint_type for b in got.bodies] == [b.joint_type for b in md.bodies] and got.num_dofs == md.num_dofs
    _same(got, md)
    # EulerJoint's flipAxisMap: R_a(f q) = R_{f a}(q) -> the chain rotates about the flipped unit axis
    _Joint._flip = (1.0, -1.0, 1.0)
    try:
        flipped = model_from_nimble_world(StandInWorld(shell), name=md.name, max_contacts=0)
    finally:
        _Joint._flip = (1.0, 1.0, 1.0)
    k = [i for i, b in enumerate(flipped.bodies) if b.name == "upper#v1"][0]
    assert tuple(flipped.bodies[k].axis) == (0.0, -1.0, 0.0) and tuple(got.bodies[k].axis) == (0.0, 1.0, 0.0)
