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


from nimblephysics_amd.loaders import load_urdf, with_ground  # noqa: E402


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


if __name__ == "__main__":
    main()
