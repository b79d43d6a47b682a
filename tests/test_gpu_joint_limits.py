"""Joint-limit constraint rows on the device (JointLimitConstraint.cpp; nbl_model_desc.dof_limit_enforced) against the CPU oracle, through
the C ABI: EVERY world's next state and both gradients (tests/parity.py), limit rows alone and in one LCP with contacts, warm-started
second steps (the cache carries the reference's sign of the upper-limit rows), the status bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _states(md, B, seed, at_limit=0.5, spread=0.25):
    """Random states; every limited DOF sits exactly at / beyond one of its limits with probability `at_limit`."""
    rng = np.random.default_rng(seed)
    fl = md.flat()
    n = md.num_dofs
    q = np.clip(rng.normal(0, spread, (B, n)), -0.3, 0.3); v = rng.normal(0, 0.6, (B, n))
    for d in range(n):
        if not fl["dof_limit_enforced"][d]:
            continue
        r = rng.random(B)
        lo, hi = fl["pos_lo"][d], fl["pos_hi"][d]
        q[r < at_limit / 2, d] = lo - rng.choice([0.0, 0.0, 0.03], (r < at_limit / 2).sum())
        sel = (r >= at_limit / 2) & (r < at_limit)
        q[sel, d] = hi + rng.choice([0.0, 0.0, 0.03], sel.sum())
    return np.concatenate([q, v], 1), rng.normal(0, 0.3, (B, len(md.action_map)))


def _compare(tag, md, s, a, seed, min_limit=0.5, min_contact=None, lcp=None):
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
    assert np.array_equal(status & 0x481, ref["status"] & 0x481), (np.unique(status & 0x481), np.unique(ref["status"] & 0x481))
    assert (status & 0x400).astype(bool).mean() >= min_limit
    if min_contact is not None:
        assert ((status & 0x401) == 0x401).mean() >= min_contact, ((status & 0x401) == 0x401).mean()      # limit rows and contacts in one LCP
    dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
    assert not (status & 0x80).any()                                             # no world runs out of constraint slots: none is left out
    assert_match_or_reference_unstable(tag, ow, s, a, g, dev, {k: ref[k] for k in dev}, TOL, max_unstable=0.02 * len(s))
    print(f"[{tag}] limit rows in {(status & 0x400).astype(bool).mean():.2f} of the worlds, contacts in {(status & 1).mean():.2f}, stage 0 resolved "
          f"{((status & 0x2) != 0).mean():.2f}, overflow {(status & 0x80).astype(bool).mean():.3f}")
    return world, ow, status


def test_limit_rows_alone():
    from util import limited_arm
    md = limited_arm()
    s, a = _states(md, 512, 1)
    _compare("limited arm", md, s, a, 2, min_limit=0.8)


def test_one_limited_joint_cartpole_both_sides_and_moving_inwards():
    import copy
    import nimblephysics_amd as na
    md = copy.deepcopy(na.cartpole())
    md.bodies[1].pos_lo, md.bodies[1].pos_hi, md.bodies[1].limit_enforced = (-0.5,), (0.5,), True
    s, a = _states(md, 256, 3, at_limit=0.9)
    world, ow, status = _compare("cartpole with a limited pole", md, s, a, 4, min_limit=0.8)
    # a pole at its limit moving outwards is stopped
    import torch
    nxt = world.from_soa(world.step_soa(world.to_soa(torch.tensor(s, device="cuda:0")), world.to_soa(torch.tensor(a, device="cuda:0")))[0]).cpu().numpy()
    outward = ((s[:, 1] >= 0.5) & (s[:, 3] > 0.05)) | ((s[:, 1] <= -0.5) & (s[:, 3] < -0.05))
    assert outward.sum() > 20 and np.abs(nxt[outward, 3]).max() < 1e-9


def test_limit_rows_and_contacts_in_one_lcp():
    from util import limited_arm
    md = limited_arm(ground=True)
    s, a = _states(md, 1024, 5, at_limit=0.35)
    # the base's box on the ground in most worlds (slide position in (-0.03, 0]: four corner contacts = 12 rows, plus the limit rows)
    rng = np.random.default_rng(6)
    s[:, 0] = rng.uniform(-0.025, 0.008, len(s))
    _compare("limited arm on the ground", md, s, a, 7, min_limit=0.5, min_contact=0.3)


def test_warm_started_second_step_uses_the_cache_in_the_references_sign():
    """Two chained steps; the oracle's second step starts from the DEVICE's first-step solution (lcp cache, the reference's sign
    convention: an upper-limit row's impulse is <= 0)."""
    import torch
    import nimblephysics_amd as na
    from oracle import OracleWorld
    from util import limited_arm
    md = limited_arm()
    s, a = _states(md, 256, 8, at_limit=0.6)
    s[:, md.num_dofs:] *= 0.05                                                   # slow: the same rows are active in the second step
    world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
    st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
    n1, _, _ = world.step_soa(st, at)
    cache = world.lcp_cache.cpu().numpy().T.copy()                                # [B, 25]
    rows = cache[:, 24].astype(int)
    fl = md.flat()
    up = np.zeros(len(s), bool)
    for d in range(md.num_dofs):
        up |= s[:, d] >= fl["pos_hi"][d]
    assert (cache[up, :24].min(1) < -1e-9).sum() > 10                             # upper-limit rows carry negative impulses in the cache
    n2, _, status2 = world.step_soa(n1, at)
    s1 = world.from_soa(n1).cpu().numpy()
    # the device keeps three row slots per constraint (a limit row has two empty tangent slots), the reference one row per limit
    compact = np.zeros((len(s), 24)); compact[:, :8] = cache[:, 0:24:3]      # (the oracle reads rows of 3 * max_contacts doubles)
    ref = ow.step_batch(s1, a, None, threads=8, lcp_in=compact, lcp_len_in=rows // 3)
    err = np.abs(world.from_soa(n2).cpu().numpy() - ref["next"]).max(1)
    st2 = status2.cpu().numpy().astype(np.uint32)
    assert np.array_equal(st2 & 0x481, ref["status"] & 0x481)
    assert ((st2 & 0x402) == 0x402).mean() > 0.3                                  # warm start: many limit worlds resolve at stage 0
    assert err.max() < 1e-9, err.max()


def test_limits_on_the_coordinates_of_ball_joints_and_of_free_joints_below_the_root():
    """JointLimitConstraint works on the generalized coordinates of ANY joint: exponential coordinates of a BallJoint, the six coordinates
    of a FreeJoint.  On the device those joints are chains of coincident single-axis bodies that carry the joint's own generalized
    velocities, so the same pseudo-contact serves (between two bodies of the chain).  Refused: limits on a free-joint ROOT."""
    import nimblephysics_amd as na
    I = (0.003, 0.004, 0.005, 0, 0, 0)
    lim = dict(pos_lo=(-0.4, -0.3, -0.5), pos_hi=(0.35, 0.45, 0.3), limit_enforced=True)
    bodies = [na.BodySpec("base", -1, "revolute", "yaw", axis=(0, 1, 0), mass=1.0, inertia=I),
              na.BodySpec("upper", 0, "ball", "shoulder", T_pj=na.make_transform((0.1, 0.2, 0)), T_cj=na.make_transform((0, 0.15, 0)), mass=0.6, inertia=I, **lim),
              na.BodySpec("lower", 1, "revolute", "elbow", axis=(1, 0, 0), T_pj=na.make_transform((0, -0.15, 0)), T_cj=na.make_transform((0, 0.12, 0)),
                          mass=0.4, inertia=I, pos_lo=(-0.2,), pos_hi=(0.6,), limit_enforced=True),
              na.BodySpec("hand", 2, "free", "wrist", T_pj=na.make_transform((0, -0.12, 0)), mass=0.2, inertia=I,
                          pos_lo=(-0.3, -0.3, -0.3, -0.05, -0.05, -0.05), pos_hi=(0.3, 0.3, 0.3, 0.05, 0.05, 0.05), limit_enforced=True)]
    md = na.ModelDescription("limited_ball_arm", bodies, [], max_contacts=16)      # 10 limited coordinates: more rows than the 24-row build has slots
    s, a = _states(md, 512, 11, at_limit=0.25, spread=0.15)
    _compare("ball / free joint limits", md, s, a, 12, min_limit=0.6)
    root = na.ModelDescription("limited_root", [na.BodySpec("b", -1, "free", "root", mass=1.0, inertia=I, pos_lo=(-1,) * 6, pos_hi=(1,) * 6, limit_enforced=True)],
                               [], max_contacts=8)
    with pytest.raises(Exception, match="free-joint root"):
        na.World(root, device="cuda:0")


def test_limit_enforcement_needs_contact_slots_and_single_dof_joints():
    import ctypes as C
    import nimblephysics_amd as na
    from nimblephysics_amd import _lib
    from util import limited_arm
    md = limited_arm()
    desc, keep = md.to_desc()
    desc.max_contacts = 0
    h = C.c_void_p()
    assert _lib.lib().nbl_model_create(C.byref(desc), 0, C.byref(h)) != 0 and b"max_contacts" in _lib.lib().nbl_last_error()
