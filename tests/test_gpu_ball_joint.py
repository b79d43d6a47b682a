"""Ball joints on the device (three coincident single-axis joints + the closed-form acceleration term, positions through H(q) of the
first one; nimble_amd.hip expandBallJoints) against the CPU oracle's real 3-DOF BallJoint restatement: next state and gradients to 1e-7,
free fall and in contact, ball joints below a free root, below a revolute root and in two branches; mass gradients and rollouts use the
caller's body indices."""
import numpy as np
import pytest

from test_ball_joint import ball_model

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _compare(md, B, seed, on_ground=False, vel=1.0):
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    rng = np.random.default_rng(seed)
    n = md.num_dofs; k = len(md.action_map)
    q = rng.normal(0, 0.5, (B, n)); v = rng.normal(0, vel, (B, n))
    if on_ground:
        q[:, 3] = rng.normal(0, 0.2, B); q[:, 5] = rng.normal(0, 0.2, B); q[:, 4] = rng.uniform(0.0, 0.2, B)
    s = np.concatenate([q, v], 1); a = rng.normal(0, 1.0, (B, k)); g = rng.normal(0, 1, s.shape)
    world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy()
    out.backward(torch.tensor(g, device="cuda:0"))
    ref = ow.step_batch(s, a, g, threads=8)
    sc = lambda x: max(np.abs(x).max(), 1e-30)
    dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
    err = np.maximum.reduce([np.abs(dev[key] - ref[key]).max(1) / sc(ref[key]) for key in dev])
    return err, status, ref["status"]


@pytest.mark.parametrize("free_root,seed", [(True, 0), (False, 1), (True, 2)])
def test_ball_joints_in_free_fall_equal_the_oracle(free_root, seed):
    md = ball_model(seed, free_root)
    err, _, _ = _compare(md, 64, 10 + seed)
    assert err.max() < TOL, err.max()


def test_large_rotations_and_fast_spins():
    """|q| up to ~pi on the ball joints (the exponential-map Jacobian far from the identity) and angular rates of tens of rad/s (the
    closed-form acceleration term is quadratic in them)."""
    md = ball_model(3, True)
    err, _, _ = _compare(md, 64, 20, vel=15.0)
    assert err.max() < TOL, err.max()


def test_ball_joints_in_contact_equal_the_oracle():
    md = ball_model(4, True, ground=True)
    err, status, rstatus = _compare(md, 256, 30, on_ground=True, vel=0.3)
    assert np.array_equal(status & 1, rstatus & 1) and (status & 1).mean() > 0.25
    ok = ((status | rstatus) & 0x80) == 0
    assert (err[ok] > 1e-5).sum() == 0 and np.median(err[ok]) < TOL, (np.sort(err[ok])[-5:], (err[ok] > TOL).sum())
