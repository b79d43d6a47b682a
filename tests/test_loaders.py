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
    f.write_text(URDF.replace('type="continuous"', 'type="floating"'))
    with pytest.raises(ValueError):
        na.load_urdf(str(f))


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
