#!/usr/bin/env python3
"""Transcribe the PARAMETERS of the reference's URDF assets into model-description JSON.

Run in the build container only (reads /root/reference/data, which does not exist on the GPU box):

    python tools/urdf_to_model.py

Writes nimblephysics_amd/data/{atlas33,atlas33_ground,atlas20,atlas20_ground}.json.  Conventions
follow dart/utils/urdf/DartLoader.cpp:
  * root link != "world"  -> FreeJoint root (:224-248); a root link named "world" is not a body (:196-222)
  * joint origin -> T_ParentBodyToJoint, T_ChildBodyToJoint = identity (:399-400)
  * inertial origin xyz -> local COM, inertia rotated by the inertial rpy (:527-545)
  * limits/damping per :403-437; `fixed` -> WeldJoint (:482-486)
  * children are visited depth-first in joint-NAME order (urdfdom keeps joints in a std::map and
    builds child_links from it), which fixes the DOF order of the skeleton
  * collision boxes -> box colliders with the collision origin as the shape's relative transform (:612-616);
    mesh colliders (libccd path, not vendored) are dropped — see SURVEY.md §7 "hard parts"
Only numeric parameters are transcribed; no reference code is copied.
"""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

REF = os.environ.get("NIMBLE_REFERENCE", "/root/reference")


from nimblephysics_amd.loaders import load_skel, load_urdf, with_ground  # noqa: E402


def main():
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nimblephysics_amd", "data")
    os.makedirs(out_dir, exist_ok=True)
    atlas_path = os.path.join(REF, "data/sdf/atlas/atlas_v3_box_colliders.urdf")
    ground_path = os.path.join(REF, "data/sdf/atlas/ground.urdf")
    ground = load_urdf(ground_path, "ground")
    arm_joints = [f"{s}_arm_{j}" for s in "lr" for j in ("shy", "shx", "ely", "elx", "wry", "wrx")]
    variants = {
        "atlas33": load_urdf(atlas_path, "atlas33"),
        # metric config: 20 DOF = free root + 12 leg revolutes + back_bkx/back_bky; arms and back_bkz welded at 0
        "atlas20": load_urdf(atlas_path, "atlas20", weld_joints=set(arm_joints + ["back_bkz"])),
    }
    for name, m in variants.items():
        for mdl in (m, with_ground(m, ground)):
            with open(os.path.join(out_dir, mdl.name + ".json"), "w") as f:
                json.dump(mdl.to_json(), f, indent=1)
            print(mdl.name, "bodies", len(mdl.bodies), "dofs", mdl.num_dofs, "boxes", len(mdl.boxes))
    # cfg1 / cfg4 from the reference's SKEL worlds (SkelParser conventions, nimblephysics_amd.loaders.load_skel)
    skel = {
        "single_pendulum": skel_models()[0],
        "box_stack": skel_models()[1],
    }
    skel.update(ball_joint_models())
    skel.update(many_collider_worlds())
    for name, mdl in skel.items():
        with open(os.path.join(out_dir, name + ".json"), "w") as f:
            json.dump(mdl.to_json(), f, indent=1)
        print(name, "bodies", len(mdl.bodies), "dofs", mdl.num_dofs, "boxes", len(mdl.boxes))


def skel_models():
    """cfg1: data/skel/test/single_pendulum.skel (its collision box dropped: one body, nothing to collide with).
    cfg4: data/skel/test/box_stacking.skel restricted to the ground and the first two cubes (8 contacts = the
    device path's row budget), bodies renamed ground / box1 / box2."""
    pend = load_skel(os.path.join(REF, "data/skel/test/single_pendulum.skel"), "single_pendulum", max_contacts=0)
    pend.boxes = []
    stack = load_skel(os.path.join(REF, "data/skel/test/box_stacking.skel"), "box_stack",
                      skeletons=["ground skeleton", "box skeleton1", "box skeleton2"], max_contacts=8)
    for b, nm in zip(stack.bodies, ("ground", "box1", "box2")):
        b.name = nm
    for b, jn in zip(stack.bodies, ("ground_joint", "box1_joint", "box2_joint")):
        b.joint_name = jn
    return pend, stack


def many_collider_worlds():
    """Three of the reference's own worlds that need the 48-row build of the library (more than 16 colliders / 32 collider pairs / 8
    contacts): data/skel/biped.skel (20 colliders), data/skel/fullbody1.skel (21 colliders: a humanoid over a ground box) and the whole
    data/skel/test/box_stacking.skel (ground + 10 cubes: 55 collider pairs), as load_skel reads them."""
    out = {}
    for nm, rel in (("biped", "data/skel/biped.skel"), ("fullbody1", "data/skel/fullbody1.skel"), ("box_stacking_full", "data/skel/test/box_stacking.skel")):
        md = load_skel(os.path.join(REF, rel), nm)
        if md.boxes:
            md.max_contacts = 16
        out[nm] = md
    return out


def ball_joint_models():
    """The reference's ball-joint test worlds (data/skel/test, used by its dynamics / joint unit tests): a serial chain of 10 ball joints
    and a 13-body tree of ball joints."""
    return {nm: load_skel(os.path.join(REF, "data/skel/test", nm + ".skel"), nm, max_contacts=0)
            for nm in ("serial_chain_ball_joint", "tree_structure_ball_joint")}


if __name__ == "__main__":
    main()
