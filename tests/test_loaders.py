"""URDF subset loader (conventions of dart/utils/urdf/DartLoader.cpp): a small hand-written URDF must
give the hand-built model, the DOF order must follow the joint-NAME order of urdfdom's std::map, fixed joints must merge
away, and (when the reference tree is present) the committed Atlas descriptions must be reproducible from its URDF."""
import json
import os

import numpy as np
import pytest

import nimblephysics_amd as na
from nimblephysics_amd.model import BodySpec, BoxSpec, ModelDescription, make_transform

URDF = """<?xml version="1.0"?>
<robot name="arm">
  <link name="base"><inertial><mass value="2.0"/><origin xyz="0 0.1 0" rpy="0 0 0"/>
    <inertia ixx="0.1" iyy="0.2" izz="0.3" ixy="0" ixz="0" iyz="0"/></inertial>
    <collision><origin xyz="0 -0.05 0" rpy="0 0 0"/><geometry><box size="0.4 0.1 0.4"/></geometry></collision></link>
  <link name="upper"><inertial><mass value="1.0"/><origin xyz="0 0 0.2" rpy="0 0 0.5"/>
    <inertia ixx="0.01" iyy="0.02" izz="0.03" ixy="0.001" ixz="0" iyz="0"/></inertial></link>
  <link name="lower"><inertial><mass value="0.5"/><origin xyz="0 0 0.15"/>
    <inertia ixx="0.005" iyy="0.005" izz="0.001" ixy="0" ixz="0" iyz="0"/></inertial></link>
  <link name="tool"><inertial><mass value="0.1"/><origin xyz="0 0 0.02"/>
    <inertia ixx="1e-4" iyy="1e-4" izz="1e-4" ixy="0" ixz="0" iyz="0"/></inertial></link>
  <link name="slider"><inertial><mass value="0.3"/><origin xyz="0 0 0"/>
    <inertia ixx="1e-3" iyy="1e-3" izz="1e-3" ixy="0" ixz="0" iyz="0"/></inertial></link>
  <joint name="z_shoulder" type="revolute"><parent link="base"/><child link="upper"/><origin xyz="0 0 0.1" rpy="0 0.3 0"/>
    <axis xyz="0 1 0"/><limit lower="-1" upper="2" effort="50" velocity="7"/><dynamics damping="0.4"/></joint>
  <joint name="elbow" type="continuous"><parent link="upper"/><child link="lower"/><origin xyz="0 0 0.4"/><axis xyz="1 0 0"/></joint>
  <joint name="wrist_fixed" type="fixed"><parent link="lower"/><child link="tool"/><origin xyz="0 0 0.3"/></joint>
  <joint name="a_slide" type="prismatic"><parent link="base"/><child link="slider"/><origin xyz="0.2 0 0"/><axis xyz="0 0 1"/>
    <limit lower="0.1" upper="0.5" effort="10" velocity="1"/></joint>
</robot>
"""


def test_urdf_subset_loader_builds_the_expected_model(tmp_path):
    f = tmp_path / "arm.urdf"
    f.write_text(URDF)
    md = na.load_urdf(str(f))
    assert md.name == "arm"
    # depth-first, children in joint-NAME order: a_slide before z_shoulder
    assert [b.name for b in md.bodies] == ["base", "slider", "upper", "lower", "tool"]
    assert [b.joint_type for b in md.bodies] == ["free", "prismatic", "revolute", "revolute", "weld"]
    assert md.num_dofs == 6 + 1 + 1 + 1
    sl, up, lo = md.bodies[1], md.bodies[2], md.bodies[3]
    assert sl.pos_lo == (0.1,) and sl.pos_hi == (0.5,) and sl.rest == (0.3,)          # rest moved inside the limits (DartLoader.cpp:414-431)
    assert up.damping == (0.4,) and up.force_hi == (50.0,) and up.vel_lo == (-7.0,) and up.axis == (0.0, 1.0, 0.0)
    assert np.allclose(up.T_pj, make_transform((0, 0, 0.1), (0, 0.3, 0)))
    # inertia rotated by the inertial rpy
    R = make_transform((0, 0, 0), (0, 0, 0.5))[:3, :3]
    J = R @ np.array([[0.01, 0.001, 0], [0.001, 0.02, 0], [0, 0, 0.03]]) @ R.T
    assert np.allclose(up.inertia, (J[0, 0], J[1, 1], J[2, 2], J[0, 1], J[0, 2], J[1, 2]))
    assert lo.pos_lo == BodySpec("x", -1, "revolute", "j").pos_lo                      # continuous: no position limits
    assert len(md.boxes) == 1 and md.boxes[0].body == 0 and md.boxes[0].size == (0.4, 0.1, 0.4)
    # the weld merges away and the merged model equals the hand-built one
    flat = md.merge_welds().flat()
    hand = ModelDescription("arm", [
        BodySpec("base", -1, "free", "rootJoint", mass=2.0, com=(0, 0.1, 0), inertia=(0.1, 0.2, 0.3, 0, 0, 0)),
        BodySpec("slider", 0, "prismatic", "a_slide", axis=(0, 0, 1), T_pj=make_transform((0.2, 0, 0)), mass=0.3, com=(0, 0, 0),
                 inertia=(1e-3, 1e-3, 1e-3, 0, 0, 0), pos_lo=(0.1,), pos_hi=(0.5,), vel_lo=(-1.0,), vel_hi=(1.0,), force_lo=(-10.0,),
                 force_hi=(10.0,), rest=(0.3,)),
        BodySpec("upper", 0, "revolute", "z_shoulder", axis=(0, 1, 0), T_pj=make_transform((0, 0, 0.1), (0, 0.3, 0)), mass=1.0,
                 com=(0, 0, 0.2), inertia=tuple(up.inertia), pos_lo=(-1.0,), pos_hi=(2.0,), vel_lo=(-7.0,), vel_hi=(7.0,),
                 force_lo=(-50.0,), force_hi=(50.0,), damping=(0.4,)),
        BodySpec("lower", 2, "revolute", "elbow", axis=(1, 0, 0), T_pj=make_transform((0, 0, 0.4)), mass=0.5, com=(0, 0, 0.15),
                 inertia=(0.005, 0.005, 0.001, 0, 0, 0)),
        BodySpec("tool", 3, "weld", "wrist_fixed", T_pj=make_transform((0, 0, 0.3)), mass=0.1, com=(0, 0, 0.02),
                 inertia=(1e-4, 1e-4, 1e-4, 0, 0, 0)),
    ], [BoxSpec(0, make_transform((0, -0.05, 0)), (0.4, 0.1, 0.4), 1.0)]).merge_welds().flat()
    assert flat.keys() == hand.keys()
    for k in flat:
        assert np.allclose(np.asarray(flat[k], dtype=float), np.asarray(hand[k], dtype=float)), k


def test_unsupported_urdf_features_raise(tmp_path):
    f = tmp_path / "bad.urdf"
    f.write_text(URDF.replace('type="continuous"', 'type="spherical"'))
    with pytest.raises(ValueError):
        na.load_urdf(str(f))
    f.write_text(URDF.replace("</robot>", '<link name="hull"><collision><geometry><mesh filename="hull.stl"/></geometry></collision></link>'
                                          '<joint name="hull_fixed" type="fixed"><parent link="base"/><child link="hull"/></joint></robot>'))
    with pytest.raises(ValueError):                      # a collider outside the analytic narrow phases is not dropped silently ...
        na.load_urdf(str(f))
    assert len(na.load_urdf(str(f), drop_unsupported_colliders=True).bodies) == 6          # ... unless asked to
    import re
    first_joint = re.search(r'<joint name="([^"]+)" type="(revolute|continuous|prismatic)">', URDF)
    f.write_text(URDF.replace(first_joint.group(0), first_joint.group(0) + f'<mimic joint="{first_joint.group(1)}" multiplier="2"/>', 1))
    with pytest.raises(ValueError, match="mimic"):       # a MIMIC actuator in the reference (DartLoader.cpp:318-380)
        na.load_urdf(str(f))


def test_urdf_floating_and_planar_joints_like_dart_loader(tmp_path):
    """DartLoader::createDartJoint (DartLoader.cpp:487-503): floating -> FreeJoint (here below another body: six coincident axes on the
    device), planar -> PlanarJoint in its default XY plane (a translational-x, translational-y, rotational-z chain)."""
    f = tmp_path / "fp.urdf"
    f.write_text(URDF.replace('type="continuous"', 'type="floating"').replace('name="a_slide" type="prismatic"', 'name="a_slide" type="planar"'))
    md = na.load_urdf(str(f))
    types = {b.name: b.joint_type for b in md.bodies}
    assert types["lower"] == "free" and md.bodies[[b.name for b in md.bodies].index("lower")].parent >= 0
    assert [types[k] for k in ("slider#v0", "slider#v1", "slider")] == ["prismatic", "prismatic", "revolute"]
    assert md.num_dofs == 6 + 3 + 1 + 6


REF_URDF = "/root/reference/data/sdf/atlas/atlas_v3_box_colliders.urdf"


@pytest.mark.skipif(not os.path.exists(REF_URDF), reason="reference tree not present (GPU box)")
def test_committed_atlas_descriptions_are_reproducible_from_the_reference_urdf():
    arms = [f"{s}_arm_{j}" for s in "lr" for j in ("shy", "shx", "ely", "elx", "wry", "wrx")]
    md = na.load_urdf(REF_URDF, "atlas20", weld_joints=set(arms + ["back_bkz"]))
    ground = na.load_urdf("/root/reference/data/sdf/atlas/ground.urdf", "ground")
    both = na.with_ground(md, ground)
    here = os.path.join(os.path.dirname(na.__file__), "data", "atlas20_ground.json")
    assert json.loads(json.dumps(both.to_json())) == json.load(open(here))


def test_urdf_sphere_collision_geometry(tmp_path):
    """<sphere radius=...> collision geometry becomes a sphere collider (SphereShape of DartLoader.cpp createShape)."""
    f = tmp_path / "ball.urdf"
    f.write_text(URDF.replace('<geometry><box size="0.4 0.1 0.4"/></geometry>', '<geometry><sphere radius="0.25"/></geometry>'))
    md = na.load_urdf(str(f))
    assert len(md.boxes) == 1 and md.boxes[0].shape == "sphere" and md.boxes[0].size == (0.25, 0.25, 0.25)
    assert md.flat()["box_shape"].tolist() == [1]
    assert ModelDescription.from_json(json.loads(json.dumps(md.to_json()))).boxes[0].shape == "sphere"


# ---- SKEL subset loader (dart/utils/SkelParser.cpp conventions) --------------------------------------------------------
SKEL = """<?xml version="1.0" ?>
<skel version="1.0">
  <world name="w">
    <physics><time_step>0.002</time_step><gravity>0 -9.81 0</gravity></physics>
    <skeleton name="floor">
      <body name="slab"><transformation>0 -0.5 0 0 0 0</transformation>
        <collision_shape><transformation>0 0 0 0 0 0</transformation><geometry><box><size>4 0.2 4</size></box></geometry></collision_shape></body>
      <joint type="weld" name="fix"><parent>world</parent><child>slab</child></joint>
    </skeleton>
    <skeleton name="arm">
      <body name="b2"><transformation>0.5 0.3 0 0 0 0</transformation>
        <inertia><mass>2</mass><offset>0.1 0 0</offset></inertia>
        <collision_shape><transformation>0.1 0 0 0 0 0.2</transformation><geometry><box><size>0.4 0.1 0.2</size></box></geometry></collision_shape></body>
      <body name="b1"><transformation>0 0.3 0 0 0 0.5</transformation>
        <inertia><mass>3</mass><moment_of_inertia><ixx>0.1</ixx><iyy>0.2</iyy><izz>0.3</izz><ixy>0.01</ixy><ixz>0</ixz><iyz>0</iyz></moment_of_inertia></inertia>
        <collision_shape><geometry><ellipsoid><size>0.2 0.2 0.2</size></ellipsoid></geometry></collision_shape></body>
      <joint type="revolute" name="j2"><parent>b1</parent><child>b2</child><transformation>-0.2 0 0 0 0 0</transformation>
        <axis><xyz>0 0 1</xyz><dynamics><damping>0.3</damping><spring_stiffness>2.5</spring_stiffness><spring_rest_position>0.1</spring_rest_position></dynamics>
          <limit><lower>-1.5</lower><upper>1.0</upper></limit></axis></joint>
      <joint type="prismatic" name="j1"><parent>world</parent><child>b1</child><axis><xyz>1 0 0</xyz><damping>0.7</damping></axis></joint>
    </skeleton>
  </world>
</skel>
"""


def test_skel_subset_loader_builds_the_expected_model(tmp_path):
    from nimblephysics_amd.loaders import euler_xyz_to_matrix
    f = tmp_path / "w.skel"
    f.write_text(SKEL)
    md = na.load_skel(str(f))
    assert md.dt == 0.002 and tuple(md.gravity) == (0.0, -9.81, 0.0)
    # parents before children, whatever the order of the <body> / <joint> elements: j1 (world -> b1) before j2 (b1 -> b2)
    assert [b.name for b in md.bodies] == ["slab", "b1", "b2"]
    assert [b.joint_type for b in md.bodies] == ["weld", "prismatic", "revolute"]
    assert [b.parent for b in md.bodies] == [-1, -1, 1]
    slab, b1, b2 = md.bodies
    Rz = euler_xyz_to_matrix((0, 0, 0.5))
    Tw1 = make_transform((0, 0.3, 0), R=Rz); Tw2 = make_transform((0.5, 0.3, 0))
    c2j = make_transform((-0.2, 0, 0))
    assert np.allclose(b1.T_pj, Tw1) and np.allclose(b1.T_cj, np.eye(4))                  # parent = world, no joint transformation
    assert np.allclose(b2.T_pj, np.linalg.inv(Tw1) @ Tw2 @ c2j) and np.allclose(b2.T_cj, c2j)   # SkelParser.cpp:1540-1552
    assert b1.damping == (0.7,) and b2.damping == (0.3,) and b2.spring == (2.5,) and b2.rest == (0.1,)
    assert b2.pos_lo == (-1.5,) and b2.pos_hi == (1.0,)
    assert b1.mass == 3.0 and tuple(b1.inertia) == (0.1, 0.2, 0.3, 0.01, 0.0, 0.0)
    # no <moment_of_inertia>: the first shape's inertia for that mass (BoxShape::computeInertia)
    assert np.allclose(b2.inertia, (2 / 12 * (0.1 ** 2 + 0.2 ** 2), 2 / 12 * (0.4 ** 2 + 0.2 ** 2), 2 / 12 * (0.4 ** 2 + 0.1 ** 2), 0, 0, 0))
    assert tuple(b2.com) == (0.1, 0.0, 0.0)
    assert (slab.mass, tuple(slab.inertia)) == (1.0, (1.0, 1.0, 1.0, 0.0, 0.0, 0.0))      # BodyNode defaults
    assert [(bx.body, bx.shape) for bx in md.boxes] == [(0, "box"), (1, "sphere"), (2, "box")]
    assert md.boxes[1].size == (0.1, 0.1, 0.1)                                              # ellipsoid diameters -> radius
    assert np.allclose(md.boxes[2].T, make_transform((0.1, 0, 0), R=euler_xyz_to_matrix((0, 0, 0.2))))
    only_arm = na.load_skel(str(f), skeletons=["arm"])
    assert [b.name for b in only_arm.bodies] == ["b1", "b2"] and len(only_arm.boxes) == 2
    bad = tmp_path / "bad.skel"
    bad.write_text(SKEL.replace('type="prismatic"', 'type="hinge2"'))
    with pytest.raises(ValueError):
        na.load_skel(str(bad))
    ball = tmp_path / "ball.skel"           # a ball joint with <dof> elements (readBallJoint + readAllDegreesOfFreedom)
    ball.write_text(SKEL.replace('type="prismatic"', 'type="ball"').replace(
        "<child>b1</child>", '<child>b1</child><dof local_index="1"><position lower="-1" upper="2"/><damping>0.3</damping></dof>'
                             '<dof local_index="2"><force lower="-5" upper="5"/><spring_stiffness>4</spring_stiffness></dof>'))
    mb = na.load_skel(str(ball))
    bb = [b for b in mb.bodies if b.name == "b1"][0]
    assert bb.joint_type == "ball" and mb.num_dofs == md.num_dofs + 2
    assert bb.pos_lo == (-np.inf, -1.0, -np.inf) and bb.pos_hi == (np.inf, 2.0, np.inf) and bb.damping == (0.0, 0.3, 0.0)
    assert bb.force_lo == (-np.inf, -np.inf, -5.0) and bb.spring == (0.0, 0.0, 4.0)


def test_skel_model_steps_like_the_hand_built_one(tmp_path):
    """The loaded model is a working model: the oracle steps it, and the mass matrix of the two-link arm equals the
    closed form of the same arm written by hand."""
    from oracle import OracleWorld
    f = tmp_path / "w.skel"
    f.write_text(SKEL)
    md = na.load_skel(str(f), skeletons=["arm"])
    ow = OracleWorld(md)
    q = np.array([0.2, -0.4])
    M = ow.mass_matrix(q)
    assert M.shape == (2, 2) and np.allclose(M, M.T) and np.all(np.linalg.eigvalsh(M) > 0)
    assert abs(M[0, 0] - (3.0 + 2.0)) < 1e-12                                              # prismatic root carries both masses
    s = np.concatenate([q, [0.1, -0.2]])
    nxt = ow.step(s, np.zeros(2))
    assert np.all(np.isfinite(nxt)) and not np.allclose(nxt, s)


REF_SKEL = "/root/reference/data/skel/test"


@pytest.mark.skipif(not os.path.isdir(REF_SKEL), reason="reference tree not present (GPU box)")
def test_committed_cfg1_cfg4_descriptions_are_reproducible_from_the_reference_skel_files():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from urdf_to_model import skel_models
    pend, stack = skel_models()
    data = os.path.join(os.path.dirname(na.__file__), "data")
    assert json.loads(json.dumps(pend.to_json())) == json.load(open(os.path.join(data, "single_pendulum.json")))
    assert json.loads(json.dumps(stack.to_json())) == json.load(open(os.path.join(data, "box_stack.json")))
    from urdf_to_model import ball_joint_models
    for nm, mdl in ball_joint_models().items():
        assert json.loads(json.dumps(mdl.to_json())) == json.load(open(os.path.join(data, nm + ".json"))), nm
        assert all(b.joint_type == "ball" for b in mdl.bodies)
    # what the files say (single_pendulum.skel / box_stacking.skel)
    p = na.single_pendulum()
    assert p.num_dofs == 1 and p.bodies[0].mass == 5.0 and tuple(p.bodies[0].inertia) == (1.0, 2.0, 3.0, 0.0, 0.0, 0.0) and tuple(p.bodies[0].damping) == (10.0,)
    assert np.allclose(p.bodies[0].T_pj, np.eye(4)) and np.allclose(p.bodies[0].T_cj, make_transform((-0.1, 0, 0)))
    bs = na.box_stack()
    assert [b.joint_type for b in bs.bodies] == ["weld", "free", "free"] and bs.num_dofs == 12 and bs.max_contacts == 8
    assert [b.mass for b in bs.bodies[1:]] == [0.1, 0.1] and np.allclose(bs.bodies[2].T_pj[:3, 3], (0, 0.2, 0))
    assert [tuple(bx.size) for bx in bs.boxes] == [(2.0, 0.01, 2.0), (0.2, 0.2, 0.2), (0.2, 0.2, 0.2)]


def test_load_model_dispatches_on_the_file_type(tmp_path):
    u = tmp_path / "a.urdf"; u.write_text(URDF)
    k = tmp_path / "w.skel"; k.write_text(SKEL)
    assert na.load_model(str(u)).num_dofs == na.load_urdf(str(u)).num_dofs
    assert [b.name for b in na.load_model(str(k)).bodies] == [b.name for b in na.load_skel(str(k)).bodies]
    with pytest.raises(ValueError):
        na.load_model(str(tmp_path / "x.sdf"))



SKEL_ORDER = """<?xml version="1.0" ?>
<skel version="1.0"><world name="w">
  <skeleton name="s">
    <body name="A"><transformation>0 0 0 0 0 0</transformation></body>
    <body name="B"><transformation>1 0 0 0 0 0</transformation>
      <inertia><mass>2</mass></inertia>
      <visualization_shape><geometry><box><size>0.3 0.2 0.1</size></box></geometry></visualization_shape>
      <collision_shape><geometry><box><size>1 1 1</size></box></geometry></collision_shape></body>
    <body name="C"><transformation>2 0 0 0 0 0</transformation></body>
    <body name="D"><transformation>3 0 0 0 0 0</transformation></body>
    <joint type="revolute" name="jD"><parent>C</parent><child>D</child><axis><xyz>0 0 1</xyz></axis></joint>
    <joint type="revolute" name="jC"><parent>B</parent><child>C</child><axis><xyz>0 0 1</xyz></axis></joint>
    <joint type="revolute" name="jB"><parent>world</parent><child>B</child><axis><xyz>0 0 1</xyz></axis></joint>
    <joint type="revolute" name="jA"><parent>world</parent><child>A</child><axis><xyz>0 0 1</xyz></axis></joint>
  </skeleton>
</world></skel>
"""


def test_skel_assembly_order_and_default_inertia_follow_the_reference(tmp_path):
    """readSkeleton (SkelParser.cpp:999-1040, getNextJointAndNodePair :753-805): the lowest remaining joint in file order is created next,
    after its missing ancestors - jD needs C needs B: B, C, D, and only then jA - which is the skeleton's body and DOF order.  A body
    with a mass but no moment takes the inertia of its FIRST shape node, and visualization shapes are read before collision shapes
    (:612-645)."""
    f = tmp_path / "o.skel"
    f.write_text(SKEL_ORDER)
    md = na.load_skel(str(f))
    assert [b.name for b in md.bodies] == ["B", "C", "D", "A"]
    assert [b.parent for b in md.bodies] == [-1, 0, 1, -1]
    B = md.bodies[0]
    assert np.allclose(B.inertia, (2 / 12 * (0.2 ** 2 + 0.1 ** 2), 2 / 12 * (0.3 ** 2 + 0.1 ** 2), 2 / 12 * (0.3 ** 2 + 0.2 ** 2), 0, 0, 0))
    loop = tmp_path / "loop.skel"
    loop.write_text(SKEL_ORDER.replace("<parent>world</parent><child>B</child>", "<parent>D</parent><child>B</child>"))
    with pytest.raises(ValueError):
        na.load_skel(str(loop))


def test_with_ground_keeps_the_skeletons_of_both_models_apart(tmp_path):
    """Both descriptions number their skeletons from 0: merged as they are, the ground's mobile bodies would share skeleton 0 with the
    model's and never collide with it (CollisionFilter.cpp:105-154).  An untagged model (one skeleton per tree) merged with a tagged one
    must neither split the tagged skeleton nor join it."""
    f = tmp_path / "w.skel"; f.write_text(SKEL)
    a = na.load_skel(str(f), skeletons=["arm"]); b = na.load_skel(str(f), skeletons=["arm"])
    assert set(a.body_skeletons()) == {1}                                   # the file's skeleton index
    both = na.with_ground(a, b)
    assert both.body_skeletons() == [0, 0, 1, 1]
    u = tmp_path / "a.urdf"; u.write_text(URDF)
    untagged = na.load_urdf(str(u))
    assert all(bd.skeleton < 0 for bd in untagged.bodies)
    mixed = na.with_ground(a, untagged)
    sk = mixed.body_skeletons()
    assert sk[:2] == [0, 0] and len(set(sk[2:])) == len({i for i, bd in enumerate(untagged.bodies) if bd.parent < 0}) and 0 not in sk[2:]
    for i, bd in enumerate(mixed.bodies[2:], start=2):                       # a tree of the untagged model stays one skeleton
        if bd.parent >= 0:
            assert sk[i] == sk[bd.parent]


# ---- capsule colliders (SkelParser.cpp:1302-1308; narrow phases DARTCollide.cpp:4183-4420) ------------------------------------
CAPSULE_SKEL = """<?xml version="1.0" ?>
<skel version="1.0">
  <world name="w">
    <physics><time_step>0.001</time_step><gravity>0 -9.81 0</gravity></physics>
    <skeleton name="ground">
      <body name="dome"><transformation>0 -2 0 0 0 0</transformation>
        <collision_shape><geometry>GROUND</geometry></collision_shape></body>
      <joint type="weld" name="fix"><parent>world</parent><child>dome</child></joint>
    </skeleton>
    <skeleton name="rod">
      <body name="rod"><transformation>0 0.05 0 0 0 0</transformation>
        <inertia><mass>2</mass></inertia>
        <collision_shape><transformation>0 0 0 1.5707963267948966 0 0</transformation>
          <geometry><capsule><height>0.4</height><radius>0.05</radius></capsule></geometry></collision_shape></body>
      <joint type="free" name="root"><parent>world</parent><child>rod</child></joint>
    </skeleton>
  </world>
</skel>
"""


def test_skel_capsule_colliders(tmp_path):
    f = tmp_path / "capsule.skel"
    f.write_text(CAPSULE_SKEL.replace("GROUND", "<sphere><radius>2</radius></sphere>"))
    md = na.load_skel(str(f))
    cap = [bx for bx in md.boxes if bx.shape == "capsule"]
    assert len(cap) == 1 and tuple(cap[0].size) == (0.05, 0.4, 0.0) and md.flat()["box_shape"].tolist() == [1, 2]
    # no <moment_of_inertia>: CapsuleShape::computeInertia for this mass (CapsuleShape.cpp:107-131)
    r, h, m = 0.05, 0.4, 2.0
    vc, vs = np.pi * r * r * h, 4 / 3 * np.pi * r ** 3
    mc, ms = m * vc / (vc + vs), m * vs / (vc + vs)
    ixx = mc * (h * h / 12 + r * r / 4) + ms * (h * h + 3 / 8 * h * r + 0.4 * r * r)
    assert np.allclose(md.bodies[1].inertia, (ixx, ixx, mc * r * r / 2 + ms * 0.4 * r * r, 0, 0, 0), rtol=1e-14)
    # a capsule that can meet a BOX: that pair is libccd's in the reference - refused, or loaded without the capsule on request
    f.write_text(CAPSULE_SKEL.replace("GROUND", "<box><size>4 4 4</size></box>"))
    with pytest.raises(ValueError, match="capsule"):
        na.load_skel(str(f))
    md = na.load_skel(str(f), drop_unsupported_colliders=True)
    assert [bx.shape for bx in md.boxes] == ["box"]
    # ... and the description itself refuses the pair, whoever built it
    md.boxes.append(na.CapsuleSpec(1, np.eye(4), 0.05, 0.4))
    with pytest.raises(ValueError, match="capsule"):
        md.flat()


@pytest.mark.parametrize("geometry, expect", [
    ("<sphere><radius>0.3</radius></sphere>", (2 / 5 * 2 * 0.09,) * 3),                                                  # SphereShape.cpp:91-100
    ("<ellipsoid><size>0.2 0.4 0.6</size></ellipsoid>", (2 / 20 * (0.16 + 0.36), 2 / 20 * (0.04 + 0.36), 2 / 20 * (0.04 + 0.16))),   # EllipsoidShape.cpp:125-140
    ("<cylinder><radius>0.1</radius><height>0.5</height></cylinder>", (2 * (0.03 + 0.25) / 12,) * 2 + (0.5 * 2 * 0.01,)),   # CylinderShape.cpp:104-113
    ("<cone><radius>0.1</radius><height>0.5</height></cone>", (3 / 20 * 2 * (0.01 + 2 / 3 * 0.25),) * 2 + (3 / 10 * 2 * 0.01,)),  # ConeShape.cpp:106-117
])
def test_skel_default_inertia_of_every_shape_kind_the_reference_reads(tmp_path, geometry, expect):
    """A body with a mass and no <moment_of_inertia> gets computeInertia(mass) of its first ShapeNode (SkelParser.cpp:620-645) - of every
    shape kind readShape knows (:1277-1316), also the ones that are visual only here."""
    f = tmp_path / "s.skel"
    f.write_text(f"""<?xml version="1.0" ?><skel version="1.0"><world name="w"><physics><time_step>0.001</time_step><gravity>0 -9.81 0</gravity></physics>
      <skeleton name="s"><body name="b"><inertia><mass>2</mass><offset>0 0 0</offset></inertia>
        <visualization_shape><transformation>0 0 0 0 0 0</transformation><geometry>{geometry}</geometry></visualization_shape></body>
      <joint type="revolute" name="j"><parent>world</parent><child>b</child><axis><xyz>0 0 1</xyz></axis></joint></skeleton></world></skel>""")
    md = na.load_skel(str(f))
    assert np.allclose(md.bodies[0].inertia, expect + (0.0, 0.0, 0.0), rtol=1e-14, atol=0)
    assert md.bodies[0].mass == 2.0


def test_skel_soft_bodies_are_refused(tmp_path):
    """A <soft_shape> makes the body a SoftBodyNode in the reference (SkelParser.cpp readSoftBodyNode): loading it as a rigid body would
    silently change the physics."""
    f = tmp_path / "soft.skel"
    f.write_text("""<?xml version="1.0" ?><skel version="1.0"><world name="w"><physics><time_step>0.001</time_step><gravity>0 -9.81 0</gravity></physics>
      <skeleton name="s"><body name="b"><inertia><mass>1</mass><offset>0 0 0</offset></inertia>
        <soft_shape><total_mass>1</total_mass><geometry><box><size>0.1 0.1 0.1</size><frags>3 3 3</frags></box></geometry><kv>500</kv><ke>0</ke><damp>5</damp></soft_shape></body>
      <joint type="free" name="j"><parent>world</parent><child>b</child></joint></skeleton></world></skel>""")
    with pytest.raises(ValueError, match="soft body"):
        na.load_skel(str(f))
