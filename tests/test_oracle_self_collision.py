"""Self-collision in the CPU oracle (Skeleton::enableSelfCollisionCheck, off by default; BodyNodeCollisionFilter::ignoresCollision,
CollisionFilter.cpp:105-154): contacts between two bodies of one skeleton, a DOF above both bodies (DofContactType::SELF_COLLISION,
DCC.cpp:116-130: the contact moves rigidly with it, no constraint force on it), gradients pinned like the reference pins its own:
VJP == J^T g with J by central differences of the step."""
import numpy as np
import pytest

from oracle import OracleWorld
from test_oracle_spheres import _fd_check
from util import folding_arm


def _state(q2, seed, q0=0.3):
    rng = np.random.default_rng(seed)
    return np.concatenate([[q0, 2.1, q2], rng.normal(0, 0.2, 3)]), rng.normal(0, 0.1, 3)


def test_colliders_of_one_skeleton_meet_only_with_the_self_collision_check_on():
    s, a = _state(1.91, 1)
    on, off = OracleWorld(folding_arm(True)), OracleWorld(folding_arm(False))
    on.step(s, a); off.step(s, a)
    assert on.last_status & 0x1 and not (off.last_status & 0x1)
    c = on.last_contacts()
    assert c.shape[0] == 1 and sorted(c[0, 8:10].astype(int).tolist()) == [0, 2]      # the first link's box and the tip: not adjacent bodies


def test_adjacent_bodies_are_skipped_unless_the_adjacent_body_check_is_on():
    # links 0 and 1 are folded far enough for their boxes to overlap next to joint 1 (2.1 rad): a contact only with the adjacent-body check
    s, a = _state(0.3, 2)
    plain, adj = OracleWorld(folding_arm(True)), OracleWorld(folding_arm(True, adjacent=True))
    plain.step(s, a); adj.step(s, a)
    pairs = lambda w: {tuple(sorted(r[8:10].astype(int).tolist())) for r in w.last_contacts()}
    assert (0, 1) not in pairs(plain)
    adj_pairs = pairs(adj)
    # (deep overlap next to the joint: beyond the clipping depth nothing is reported; the pair is at least TESTED - see the GPU pair count)
    assert (0, 1) in adj_pairs or len(adj_pairs) == 0


@pytest.mark.parametrize("tip,q2", [("sphere", 1.905), ("box", 1.8825), ("box", 1.9725), ("box", 2.0325)])
def test_self_contact_gradient_with_a_dof_above_both_bodies(tip, q2):
    md = folding_arm(True, tip)
    s, a = _state(q2, 3)
    w = _fd_check(md, s, a, 4, tol=2e-5)
    c = w.last_contacts()
    assert c.shape[0] >= 1 and all(sorted(r[8:10].astype(int).tolist()) == [0, 2] for r in c)
    # joint 0 is above both bodies: its column of A_c is zero (getControlForceMultiple: 0), i.e. a rigid motion of the whole arm changes nothing
    # (gravity aside): the step Jacobian's rows of the relative coordinates do not depend on q0 through the contact
    g = np.zeros(6); g[4] = 1.0
    w.step(s, a); gs1, _ = w.backprop(g)
    md0 = folding_arm(True, tip); md0.gravity = (0.0, 0.0, 0.0)
    w0 = OracleWorld(md0); w0.step(s, a); gs0, _ = w0.backprop(g)
    assert abs(gs0[0]) < 1e-9                                                           # without gravity: exactly nothing


def test_a_child_on_a_compound_joint_is_adjacent_to_its_real_parent_not_to_a_virtual_link():
    """BodyNodeCollisionFilter::areAdjacentBodies compares BodyNode::getParentBodyNode (CollisionFilter.cpp:150-154).  A universal / Euler /
    planar joint is expanded here into a chain of 1-DOF joints through massless virtual links ('#v*'): with self-collision on and the
    adjacent-body check off the pair (child on a universal joint, its real parent) must stay untested, exactly like the same child on a
    revolute joint - in the description's filter, in the arrays handed to the device (box_node / box_node_parent) and in the oracle's
    narrow phase (the virtual link was taken for the parent and the pair was tested)."""
    import nimblephysics_amd as na
    from oracle import OracleWorld
    I = (0.002, 0.002, 0.002, 0, 0, 0)
    for child_joint, kw in (("revolute", dict(axis=(0, 0, 1))), ("universal", dict(axes=[(0, 0, 1), (1, 0, 0)])), ("euler_xyz", {})):
        for adjacent_check in (False, True):
            opts = dict(self_collision=True, adjacent_body_check=adjacent_check)
            bodies = [na.BodySpec("l0", -1, "revolute", "j0", axis=(0, 0, 1), mass=1.0, inertia=I, **opts),
                      na.BodySpec("l1", 0, child_joint, "j1", T_pj=na.make_transform((0.1, 0, 0)), mass=0.8, inertia=I, **opts, **kw)]
            cols = [na.BoxSpec(0, np.eye(4), (0.3, 0.1, 0.1), 0.8), na.BoxSpec(1, na.make_transform((0, 0.09, 0)), (0.3, 0.1, 0.1), 0.8)]   # 1 cm deep in one another
            md = na.ModelDescription("pair", bodies, cols, gravity=(0, -9.81, 0), max_contacts=8)
            tested = md.colliders_are_tested(md.boxes[0], md.boxes[1])
            assert tested == adjacent_check, (child_joint, adjacent_check)
            fl = md.flat()
            adj = fl["box_node_parent"][1] == fl["box_node"][0] or fl["box_node_parent"][0] == fl["box_node"][1]
            assert adj, (child_joint, fl["box_node"], fl["box_node_parent"])
            ow = OracleWorld(md)
            n = md.num_dofs
            ow.step(np.zeros(2 * n), np.zeros(n))
            assert bool(ow.last_status & 0x1) == adjacent_check, (child_joint, adjacent_check, hex(ow.last_status))
