"""World.getStateJacobian / getActionJacobian (World.cpp:2210-2243, tested upstream by
unit/test_RL_API.cpp:146-189 against finite differences): dense Jacobians assembled from 2n vector-Jacobian products,
checked against central finite differences of the GPU step itself and against J^T g = backward(g)."""
import numpy as np
import pytest

from util import contact_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("contact", [False, True])
def test_state_and_action_jacobians_vs_finite_differences(contact):
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    B = 8
    md, s0, a0 = contact_inputs("atlas20", B, 51)
    if not contact:
        md = na.atlas("atlas20", ground=False)
    world = na.World(md, device="cuda:0")
    n2, k = s0.shape[1], a0.shape[1]
    st = torch.tensor(s0, device="cuda:0", requires_grad=True)
    at = torch.tensor(a0, device="cuda:0", requires_grad=True)
    g = np.random.default_rng(1).normal(0, 1, s0.shape)
    out = timestep(world, st, at)
    JS = world.getStateJacobian().cpu().numpy()          # [B, 2n, 2n]
    JA = world.getActionJacobian().cpu().numpy()         # [B, 2n, k]
    out.backward(torch.tensor(g, device="cuda:0"))
    assert JS.shape == (B, n2, n2) and JA.shape == (B, n2, k)
    # J^T g == backward(g)
    assert np.allclose(np.einsum("bij,bi->bj", JS, g), st.grad.cpu().numpy(), rtol=1e-9, atol=1e-9 * np.abs(st.grad.cpu().numpy()).max())
    assert np.allclose(np.einsum("bij,bi->bj", JA, g), at.grad.cpu().numpy(), rtol=1e-9, atol=1e-9 * max(np.abs(at.grad.cpu().numpy()).max(), 1e-30))

    def f(s, a):
        w2 = na.World(md, device="cuda:0")                # cold LCP start like the differentiated step
        return timestep(w2, torch.tensor(s, device="cuda:0"), torch.tensor(a, device="cuda:0")).cpu().numpy()

    eps = 1e-6
    cols = list(range(6, n2, 5)) if contact else list(range(0, n2, 3))   # with contact the free-joint rotation columns sit on LCP kinks less often for joint DOFs
    for j in cols:
        d = np.zeros_like(s0); d[:, j] = eps
        fd = (f(s0 + d, a0) - f(s0 - d, a0)) / (2 * eps)
        scale = max(np.abs(JS[:, :, j]).max(), 1.0)
        assert np.abs(fd - JS[:, :, j]).max() <= (2e-5 if contact else 2e-6) * scale, (j, np.abs(fd - JS[:, :, j]).max(), scale)
    for j in range(0, k, 4):
        d = np.zeros_like(a0); d[:, j] = eps
        fd = (f(s0, a0 + d) - f(s0, a0 - d)) / (2 * eps)
        scale = max(np.abs(JA[:, :, j]).max(), 1e-3)
        assert np.abs(fd - JA[:, :, j]).max() <= 2e-5 * scale


@pytest.mark.parametrize("case", ["cartpole", "atlas20_freefall", "atlas20_contact", "atlas20_contact_noisy", "box_stack", "partial_action_space"])
def test_state_and_action_jacobians_equal_the_oracles_dense_jacobians(case):
    """The parity test proper (the reference's World::getStateJacobian / getActionJacobian, World.cpp:2210-2243, restated by
    the oracle from its five dense blocks posPos / velPos / posVel / velVel / forceVel exactly like BackpropSnapshot forms them):
    every entry of the device's [B, 2n, 2n] and [B, 2n, k] Jacobians against the oracle's, world by world, 1e-7 relative."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    from util import box_stack_inputs, cfg_inputs
    B = 16
    if case == "cartpole":
        md, s, a = cfg_inputs("cartpole", B, 61)
    elif case == "atlas20_freefall":
        md, s, a = cfg_inputs("atlas20", B, 62)
    elif case == "atlas20_contact":
        md, s, a = contact_inputs("atlas20", B, 63)
    elif case == "atlas20_contact_noisy":       # half of the worlds go through the LCP cascade
        md, s, a = contact_inputs("atlas20", B, 64, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
    elif case == "box_stack":
        md, s, a = box_stack_inputs(B, 65)
    else:
        md, s, a = cfg_inputs("atlas20", B, 66)
        md.set_action_space([6, 9, 12, 19])
        a = a[:, :4]
    world = na.World(md, device="cuda:0")
    out = timestep(world, torch.tensor(s, device="cuda:0"), torch.tensor(a, device="cuda:0"))
    JS = world.getStateJacobian().cpu().numpy()
    JA = world.getActionJacobian().cpu().numpy()
    ow = OracleWorld(md)
    worst_s = worst_a = 0.0
    for b in range(B):
        ow.reset_lcp_cache()
        nxt = ow.step(s[b], a[b])
        assert np.abs(nxt - out[b].cpu().numpy()).max() <= 1e-7 * max(np.abs(nxt).max(), 1.0)
        RS, RA = ow.getStateJacobian(), ow.getActionJacobian()
        assert RS.shape == JS[b].shape and RA.shape == JA[b].shape
        es = np.abs(JS[b] - RS).max() / max(np.abs(RS).max(), 1.0)
        ea = np.abs(JA[b] - RA).max() / max(np.abs(RA).max(), 1e-30)
        worst_s, worst_a = max(worst_s, es), max(worst_a, ea)
    # free-joint position blocks: the oracle restates the reference's central differences (FreeJoint.cpp:950-1007), the
    # device uses the exact expression; they agree to ~1e-9, everything else to round-off
    tol = 1e-7 if case != "box_stack" else 1e-5      # cubes with yaw ~ U(-pi, pi): d logMap near |yaw| = pi amplifies the FD error
    assert worst_s < tol and worst_a < tol, (case, worst_s, worst_a)


def test_jacobians_of_a_state_on_its_limits_are_not_clipped():
    """backprop() ends with clipLossGradientsToBounds (BackpropSnapshot.cpp:425-479: with a coordinate exactly on a limit the gradient
    entry that points out of the box is zeroed); getStateJacobian / getActionJacobian (World.cpp:2210-2243) are assembled WITHOUT it.
    The device forms its Jacobians from vector-Jacobian products: positions, velocities and torques exactly on their limits in half of
    the worlds, with and without enforced joint limits - every entry against the oracle's dense Jacobians, and the clipping is still
    there in the products themselves (found by the Jacobian soak on the mixed-feature models, round 3)."""
    import os
    import sys
    import torch
    import nimblephysics_amd as na
    from oracle import OracleWorld
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak_parity
    import soak_stress
    checked = clipped = 0
    for mode, seed in (("atlimit", 5101), ("atlimit", 5102), ("limits", 5103), ("limits", 5104)):
        md, s, a, g = soak_stress.mutator(mode)(seed, *soak_parity.make_case(seed, 8, False, False, True, False))
        world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
        world.setState(torch.tensor(s)); world.setAction(torch.tensor(a))
        snap = na.neural.forwardPass(world, idempotent=True)
        st = snap.getStatus().cpu().numpy().astype(np.uint32)
        Js, Ja = snap.getStateJacobian(world).cpu().numpy(), snap.getActionJacobian(world).cpu().numpy()
        lg = snap.backpropState(world, torch.tensor(g))
        gs = lg.lossWrtState.cpu().numpy()
        for b in range(len(s)):
            ow.reset_lcp_cache(); ow.step(s[b], a[b])
            if (st[b] | ow.last_status) & 0x80:
                continue
            Rs, Ra = ow.getStateJacobian(), ow.getActionJacobian()
            assert np.abs(Js[b] - Rs).max() <= 1e-7 * max(np.abs(Rs).max(), 1e-30), (mode, seed, b, hex(st[b]))
            assert np.abs(Ja[b] - Ra).max() <= 1e-7 * max(np.abs(Ra).max(), 1e-30), (mode, seed, b, hex(st[b]))
            ogs, _ = ow.backprop(g[b])
            assert np.abs(gs[b] - ogs).max() <= 1e-7 * max(np.abs(ogs).max(), 1e-30)
            clipped += int(np.abs(Js[b].T @ g[b] - gs[b]).max() > 1e-6 * np.abs(gs[b]).max())
            checked += 1
    assert checked >= 24 and clipped >= 4, (checked, clipped)
