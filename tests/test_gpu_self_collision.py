"""Self-collision on the device (nbl_model_desc.body_self_collision = Skeleton::enableSelfCollisionCheck / enableAdjacentBodyCheck; off by
default like in the reference) against the CPU oracle, through the C ABI: contacts between two bodies of one skeleton with a DOF above both
of them - EVERY world's next state and both gradients (tests/parity.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _states(B, seed, q2_lo, q2_hi):
    rng = np.random.default_rng(seed)
    q = np.stack([rng.uniform(-1.0, 1.0, B), 2.1 + rng.normal(0, 0.01, B), rng.uniform(q2_lo, q2_hi, B)], 1)
    return np.concatenate([q, rng.normal(0, 0.3, (B, 3))], 1), rng.normal(0, 0.2, (B, 3))


def _compare(tag, md, s, a, seed, min_contact):
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    from parity import assert_match_or_reference_unstable
    world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
    g = np.random.default_rng(seed).normal(0, 1, s.shape)
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy().astype(np.uint32)
    out.backward(torch.tensor(g, device="cuda:0"))
    ref = ow.step_batch(s, a, g, threads=8)
    assert np.array_equal(status & 0x81, ref["status"] & 0x81)
    assert (status & 1).mean() >= min_contact, (status & 1).mean()
    dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
    assert_match_or_reference_unstable(tag, ow, s, a, g, dev, ref, TOL, max_unstable=0.02 * len(s))
    return status


@pytest.mark.parametrize("tip,lo,hi", [("sphere", 1.895, 1.95), ("box", 1.872, 2.14)])
def test_folding_arm_touches_its_own_first_link(tip, lo, hi):
    from util import folding_arm
    md = folding_arm(True, tip)
    s, a = _states(512, 1, lo, hi)
    _compare(f"folding arm, {tip} tip on link 0", md, s, a, 2, 0.7)


def test_without_the_flag_the_same_states_are_free_motion():
    import torch
    import nimblephysics_amd as na
    from util import folding_arm
    s, a = _states(64, 3, 1.9, 1.95)
    w = na.World(folding_arm(False), device="cuda:0")
    w.step_soa(w.to_soa(torch.tensor(s, device="cuda:0")), w.to_soa(torch.tensor(a, device="cuda:0")))
    assert not (w.last_status.cpu().numpy() & 1).any()


def _two_skeleton_model(second):
    """A folding arm (self-collision on) plus a second skeleton: another arm 1 m away (two constrained groups), or a free ball (one group once
    it touches the arm)."""
    import copy
    import nimblephysics_amd as na
    from util import folding_arm
    a0 = folding_arm(True, "sphere")
    bodies, boxes = copy.deepcopy(a0.bodies), copy.deepcopy(a0.boxes)
    for b in bodies:
        b.skeleton = 0
    if second == "arm":
        for b in copy.deepcopy(a0.bodies):
            b.name += "_1"; b.joint_name += "_1"; b.parent = b.parent if b.parent < 0 else b.parent + 3
            if b.parent < 0:
                b.T_pj = na.make_transform((0, 0, 1.0))
            b.skeleton = 1; b.mass *= 1.3
            bodies.append(b)
        for bx in copy.deepcopy(a0.boxes):
            bx.body += 3
            boxes.append(bx)
    else:
        bodies.append(na.BodySpec("ball", -1, "free", "ball_joint", mass=0.3, inertia=(2e-4, 2e-4, 2e-4, 0, 0, 0), skeleton=1))
        boxes.append(na.SphereSpec(3, np.eye(4), 0.05, 0.7))
    return na.ModelDescription("arm_and_" + second, bodies, boxes, gravity=(0, -9.81, 0), max_contacts=8)


def test_two_arms_each_touching_itself_two_constrained_groups():
    md = _two_skeleton_model("arm")
    B = 256
    s1, a1 = _states(B, 6, 1.895, 1.94); s2, a2 = _states(B, 7, 1.895, 1.94)
    s = np.concatenate([s1[:, :3], s2[:, :3], s1[:, 3:], s2[:, 3:]], 1); a = np.concatenate([a1, a2], 1)
    status = _compare("two folding arms, two groups", md, s, a, 8, 0.7)
    assert not (status & 0x80).any()


def test_a_ball_resting_on_the_arm_that_touches_itself_one_group():
    """The ball's contact with link 0 unites the two skeletons: the self-contact (a DOF above both of its bodies) and a contact between
    skeletons in ONE LCP."""
    md = _two_skeleton_model("ball")
    B = 256
    s1, a1 = _states(B, 9, 1.895, 1.94)
    rng = np.random.default_rng(10)
    q0 = s1[:, 0]
    n = md.num_dofs
    s = np.zeros((B, 2 * n)); a = np.zeros((B, n))
    s[:, :3] = s1[:, :3]; s[:, n:n + 3] = s1[:, 3:]; a[:, :3] = a1
    # link 0: centre 0.15 along its x axis from the origin; the ball on its +y face, 1 - 4 mm inside, somewhere along the link
    along = rng.uniform(0.05, 0.2, B)
    ex = np.stack([np.cos(q0), np.sin(q0), 0 * q0], 1); ey = np.stack([-np.sin(q0), np.cos(q0), 0 * q0], 1)
    centre = ex * along[:, None] + ey * (0.03 + 0.05 - rng.uniform(1e-3, 4e-3, B))[:, None] + np.array([0, 0, 1.0])[None] * rng.uniform(-0.01, 0.01, B)[:, None]
    s[:, 3:6] = rng.normal(0, 0.3, (B, 3)); s[:, 6:9] = centre
    s[:, n + 3:n + 9] = rng.normal(0, 0.1, (B, 6))
    status = _compare("arm touching itself with a ball on it", md, s, a, 11, 0.7)
    assert not (status & 0x80).any()
