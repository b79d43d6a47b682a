"""The reference World's state layout when it has coordinates the device model does not: a SKEL world with an IMMOBILE
skeleton keeps that skeleton's coordinates in `World::getState()` (dart/simulation/World.cpp:2016-2047); the loader welds the skeleton at its
zero configuration and records the reference's coordinate order (ModelDescription.ref_dof_mobile), and the drop-in surface speaks the
reference's layout through nimblephysics_amd/ref_layout.py.  CPU half: the loader's bookkeeping and the layout arithmetic."""
import numpy as np
import pytest
import torch

SKEL = """<?xml version="1.0" ?>
<skel version="1.0">
  <world name="w">
    <physics><time_step>0.001</time_step><gravity>0 -9.81 0</gravity></physics>
    <skeleton name="ground skeleton">
      <mobile>false</mobile>
      <body name="ground">
        <transformation>0 -0.5 0 0 0 0</transformation>
        <inertia><mass>1</mass><offset>0 0 0</offset></inertia>
        <collision_shape><transformation>0 0 0 0 0 0</transformation><geometry><box><size>5 1 5</size></box></geometry></collision_shape>
      </body>
      <joint type="free" name="ground_joint"><parent>world</parent><child>ground</child></joint>
    </skeleton>
    <skeleton name="cube">
      <body name="box">
        <transformation>0 0.09 0 0 0 0</transformation>
        <inertia><mass>0.5</mass><offset>0 0 0</offset></inertia>
        <collision_shape><transformation>0 0 0 0 0 0</transformation><geometry><box><size>0.2 0.2 0.2</size></box></geometry></collision_shape>
      </body>
      <joint type="free" name="box_joint"><parent>world</parent><child>box</child></joint>
    </skeleton>
  </world>
</skel>
"""


def load(tmp_path):
    import warnings
    from nimblephysics_amd.loaders import load_skel
    p = tmp_path / "immobile_ground.skel"
    p.write_text(SKEL)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return load_skel(str(p))


def test_the_loader_records_the_references_coordinate_order(tmp_path):
    md = load(tmp_path)
    assert md.num_dofs == 6                                   # the device model: the cube's free joint
    assert md.ref_dof_mobile == [False] * 6 + [True] * 6     # the reference: the immobile ground's free joint first (file order)
    assert md.welded_dofs == [("ground skeleton", "ground_joint", 6)]


def test_layout_arithmetic():
    from nimblephysics_amd.ref_layout import RefLayout
    lay = RefLayout([False, True, True, False, True])
    assert (lay.n_ref, lay.n_dev) == (5, 3) and lay.dev_of == [-1, 0, 1, -1, 2]
    s = torch.arange(20, dtype=torch.float64).reshape(2, 10)
    s[:, [0, 3]] = 0.0                                        # frozen positions sit at zero
    r = lay.restrict_state(s)
    assert torch.equal(r, s[:, [1, 2, 4, 6, 7, 9]])
    with pytest.raises(ValueError, match="not zero"):
        lay.restrict_state(torch.ones(10, dtype=torch.float64))
    with pytest.raises(ValueError, match="expected 10"):
        lay.restrict_state(torch.zeros(6, dtype=torch.float64))
    full = lay.expand_state(r + 100.0, s)
    assert torch.equal(full[:, [1, 2, 4, 6, 7, 9]], r + 100.0) and torch.equal(full[:, [0, 3, 5, 8]], s[:, [0, 3, 5, 8]])
    # differentiable in both: identity on the frozen entries
    a = s.clone().requires_grad_(True)
    out = lay.expand_state(2.0 * lay.restrict_state(a), a)
    out.sum().backward()
    g = a.grad[0]
    assert torch.equal(g[[0, 3, 5, 8]], torch.ones(4, dtype=torch.float64)) and torch.equal(g[[1, 2, 4, 6, 7, 9]], 2.0 * torch.ones(6, dtype=torch.float64))
    J = lay.state_jacobian(torch.full((1, 6, 6), 7.0, dtype=torch.float64))[0]
    assert torch.equal(J[[0, 3, 5, 8]][:, [0, 3, 5, 8]], torch.eye(4, dtype=torch.float64))
    assert float(J[1, 2]) == 7.0 and float(J[0, 1]) == 0.0 and float(J[1, 0]) == 0.0
    assert lay.device_action_map([0, 1, 2, 4]) == [0, 1, 2] and lay.action_columns([0, 1, 2, 4]) == [1, 2, 3]


REF_SKEL = "/root/reference/data/skel"
JOINT_DOFS = {"free": 6, "ball": 3, "euler": 3, "revolute": 1, "prismatic": 1, "screw": 1, "universal": 2, "translational": 3, "translational2d": 2, "planar": 3, "weld": 0}


@pytest.mark.skipif(not __import__("os").path.isdir(REF_SKEL), reason="the reference's data files are not here")
@pytest.mark.parametrize("name", ["cubes.skel", "fullbody1.skel", "cartpole.skel", "ground.skel", "two_cubes.skel", "test/box_stacking.skel", "test/file_info_world_test.skel"])
def test_the_state_length_of_the_references_own_skel_files(name):
    """getStateSize() of the drop-in surface = 2 x the coordinates of EVERY skeleton of the file, immobile ones included - what
    World::getNumDofs() counts in the reference (World.cpp:2016-2047): the joints of the file, counted here straight from the XML, against
    the loader's record of the reference's coordinate order (ref_dof_mobile: one entry per reference coordinate; True = a device coordinate)."""
    import os
    import warnings
    import xml.etree.ElementTree as ET
    from nimblephysics_amd.loaders import load_skel
    from nimblephysics_amd.ref_layout import RefLayout
    path = os.path.join(REF_SKEL, name)
    want = 0
    immobile = 0
    for sk in ET.parse(path).getroot().iter("skeleton"):
        mob = (sk.findtext("mobile") or "true").strip().lower() not in ("false", "0")
        for j in sk.findall("joint"):
            nd = JOINT_DOFS[j.get("type").lower()]
            want += nd
            immobile += 0 if mob else nd
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        md = load_skel(path)
    mobile = md.ref_dof_mobile if md.ref_dof_mobile is not None else [True] * md.num_dofs
    lay = RefLayout(mobile)
    assert lay.n_ref == want, (name, lay.n_ref, want)                     # the reference's getNumDofs()
    assert lay.n_dev == md.num_dofs == want - immobile, (name, lay.n_dev, md.num_dofs, want, immobile)
    print(f"{name}: {want} reference coordinates, {immobile} of immobile skeletons (frozen), {md.num_dofs} on the device")
