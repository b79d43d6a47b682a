"""T-step rollout on the device (SingleShot::getStates / backpropGradientWrt,
dart/trajectory/SingleShot.cpp:539-700): states and gradients must equal the chain of single `timestep` calls through
torch autograd (same kernels, so bit-exact forward, gradient to round-off), and the oracle's T-step backprop."""
import numpy as np
import pytest

from util import contact_inputs

pytestmark = pytest.mark.gpu


def _chain(world, s0, acts, warm):
    import torch
    from nimblephysics_amd.timestep import timestep
    world.reset_lcp_cache()
    st = torch.tensor(s0, device="cuda:0", requires_grad=True)
    at = torch.tensor(acts, device="cuda:0", requires_grad=True)
    xs = [st]
    for t in range(acts.shape[1]):
        if not warm:
            world.reset_lcp_cache()
        xs.append(timestep(world, xs[-1], at[:, t]))
    return st, at, torch.stack(xs, 1)


@pytest.mark.parametrize("name,contact,warm", [("atlas20", True, True), ("atlas20", True, False), ("atlas20", False, True)])
def test_rollout_equals_chain_of_timesteps(name, contact, warm):
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import rollout
    B, T = 256, 6
    md, s0, a0 = contact_inputs(name, B, 21 if contact else 22)
    if not contact:
        md = na.atlas(name, ground=False)          # same pose, no colliders: free fall
    rng = np.random.default_rng(3)
    acts = np.repeat(a0[:, None, :], T, 1) + rng.normal(0, 0.05, (B, T, a0.shape[1]))
    w = rng.normal(0, 1, (B, T + 1, s0.shape[1]))          # a loss that looks at every state
    world = na.World(md, device="cuda:0")
    st, at, xs = _chain(world, s0, acts, warm)
    (xs * torch.tensor(w, device="cuda:0")).sum().backward()
    world2 = na.World(md, device="cuda:0")
    st2 = torch.tensor(s0, device="cuda:0", requires_grad=True)
    at2 = torch.tensor(acts, device="cuda:0", requires_grad=True)
    ys = rollout(world2, st2, at2, warm_start=warm)
    (ys * torch.tensor(w, device="cuda:0")).sum().backward()
    assert ys.shape == (B, T + 1, s0.shape[1])
    assert torch.equal(ys, xs.detach())
    for a, b in ((st2.grad, st.grad), (at2.grad, at.grad)):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 1e-12 * max(scale, 1.0)
    if contact:
        assert np.all(world2.rollout_status.cpu().numpy() & 0x1)


def test_rollout_vs_oracle_T_steps():
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import rollout
    from oracle import OracleWorld
    B, T = 64, 4
    md, s0, a0 = contact_inputs("atlas20", B, 31)
    acts = np.repeat(a0[:, None, :], T, 1)
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s0, device="cuda:0", requires_grad=True)
    at = torch.tensor(acts, device="cuda:0", requires_grad=True)
    ys = rollout(world, st, at, warm_start=False)
    (ys[:, -1] ** 2).sum().backward()                      # cfg5 loss: |q_T|^2 + |v_T|^2
    ow = OracleWorld(md)
    # oracle: chain of single steps, cotangent propagated by hand
    xs = [s0]
    for t in range(T):
        xs.append(ow.step_batch(xs[-1], acts[:, t], np.zeros_like(s0), threads=4)["next"])
    g = 2.0 * xs[-1]
    gas = []
    for t in range(T - 1, -1, -1):
        r = ow.step_batch(xs[t], acts[:, t], g, threads=4)
        g = r["grad_state"]; gas.append(r["grad_action"])
    gas = np.stack(gas[::-1], 1)
    assert np.abs(ys.detach().cpu().numpy()[:, -1] - xs[-1]).max() <= 1e-7 * np.abs(xs[-1]).max()
    assert np.abs(st.grad.cpu().numpy() - g).max() <= 1e-6 * np.abs(g).max()
    assert np.abs(at.grad.cpu().numpy() - gas).max() <= 1e-6 * max(np.abs(gas).max(), 1e-30)


def test_cfg5_length_rollout_T64_vs_oracle():
    """cfg5's trajectory length (T = 64, loss = |q_T|^2 + |v_T|^2) on a small batch: 64 chained GPU steps and 64 chained
    backward steps against the oracle's chain.  Errors of a chaotic contact trajectory grow with T, hence 1e-5 here."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import rollout
    from oracle import OracleWorld
    B, T = 16, 64
    md, s0, a0 = contact_inputs("atlas20", B, 33)
    acts = np.repeat(a0[:, None, :], T, 1)
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s0, device="cuda:0", requires_grad=True)
    at = torch.tensor(acts, device="cuda:0", requires_grad=True)
    ys = rollout(world, st, at, warm_start=False)
    (ys[:, -1] ** 2).sum().backward()
    assert np.all(world.rollout_status.cpu().numpy() & 0x1)           # in contact all the way
    ow = OracleWorld(md)
    xs = [s0]
    for t in range(T):
        xs.append(ow.step_batch(xs[-1], acts[:, t], np.zeros_like(s0), threads=4)["next"])
    g = 2.0 * xs[-1]
    gas = []
    for t in range(T - 1, -1, -1):
        r = ow.step_batch(xs[t], acts[:, t], g, threads=4)
        g = r["grad_state"]; gas.append(r["grad_action"])
    gas = np.stack(gas[::-1], 1)
    assert np.abs(ys.detach().cpu().numpy()[:, -1] - xs[-1]).max() <= 1e-6 * np.abs(xs[-1]).max()
    assert np.abs(st.grad.cpu().numpy() - g).max() <= 1e-5 * np.abs(g).max()
    assert np.abs(at.grad.cpu().numpy() - gas).max() <= 1e-5 * max(np.abs(gas).max(), 1e-30)


def test_cfg5_full_size_rollout_properties():
    """cfg5 at its per-GPU size: Atlas-33 on the ground, B = 8192 (one GPU's share of 65536), T = 64, loss |q_T|^2 + |v_T|^2.
    The oracle cannot run this in seconds, so size-independent properties: two runs are bit-identical; the first 64 worlds
    of the big batch equal a separate 64-world rollout bit for bit (batch independence through all 64 steps and the whole
    backward chain); the backward pass is linear in the loss scale; everything is finite and stays in contact."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import rollout
    B, T = 8192, 64
    md, s0, a0 = contact_inputs("atlas33", B, 5, joint_noise=0.02, vel_noise=0.01, action_noise=0.1)      # cfg5's own distribution

    def run(s, a, scale=1.0):
        world = na.World(md, device="cuda:0")
        st = torch.tensor(s, device="cuda:0", requires_grad=True)
        at = torch.tensor(np.repeat(a[:, None, :], T, 1), device="cuda:0", requires_grad=True)
        ys = rollout(world, st, at, warm_start=True)
        (scale * (ys[:, -1] ** 2).sum()).backward()
        return ys[:, -1].detach().cpu().numpy(), st.grad.cpu().numpy(), at.grad.cpu().numpy(), world.rollout_status.cpu().numpy()

    y1, gs1, ga1, status = run(s0, a0)
    assert np.isfinite(y1).all() and np.isfinite(gs1).all() and np.isfinite(ga1).all()
    assert np.all(status & 0x1)
    y2, gs2, ga2, _ = run(s0, a0)
    assert np.array_equal(y1, y2) and np.array_equal(gs1, gs2) and np.array_equal(ga1, ga2)
    ys, gss, gas, _ = run(s0[:64], a0[:64])
    assert np.array_equal(y1[:64], ys) and np.array_equal(gs1[:64], gss) and np.array_equal(ga1[:64], gas)
    y3, gs3, ga3, _ = run(s0[:512], a0[:512], scale=-2.0)
    assert np.abs(gs3 + 2.0 * gs1[:512]).max() <= 1e-12 * np.abs(gs1).max()
    assert np.abs(ga3 + 2.0 * ga1[:512]).max() <= 1e-12 * np.abs(ga1).max()


def test_hip_graph_replay_equals_the_eager_step():
    """GraphedStep captures step_soa + backward_soa into one HIP graph; replaying it with new inputs written into the static
    tensors must reproduce the eager results bit for bit (with and without contact)."""
    import torch
    import nimblephysics_amd as na
    from util import cfg_inputs
    for md, s, a in (contact_inputs("atlas20", 512, 61), cfg_inputs("cartpole", 512, 62)):
        world = na.World(md, device="cuda:0")
        gs = na.GraphedStep(world, 512).capture()
        for trial in range(3):
            rng = np.random.default_rng(70 + trial)
            s2 = s + rng.normal(0, 1e-3, s.shape); g = rng.normal(0, 1, s.shape)
            st = world.to_soa(torch.tensor(s2, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0")); gt = world.to_soa(torch.tensor(g, device="cuda:0"))
            gs.state.copy_(st); gs.action.copy_(at); gs.grad_next.copy_(gt)
            nxt, dstate, daction = gs.replay()
            torch.cuda.synchronize()
            ref = na.World(md, device="cuda:0")
            n2, sv, _ = ref.step_soa(st, at)
            d2, a2 = ref.backward_soa(sv, gt)
            assert torch.equal(nxt, n2) and torch.equal(dstate, d2) and torch.equal(daction, a2)


def test_hip_graph_replay_of_a_rollout_equals_the_eager_rollout():
    """GraphedRollout: the T-step rollout forward + backward (every slice stream of the library forked from and joined to the capturing
    stream) as ONE HIP graph; replays with new values in the static tensors reproduce the eager entry points bit for bit."""
    import torch
    import nimblephysics_amd as na
    md, s, a = contact_inputs("atlas20", 2048, 63)         # 2048 worlds: two slices inside the call
    B, T = s.shape[0], 6
    world = na.World(md, device="cuda:0")
    gr = na.GraphedRollout(world, B, T).capture()
    for trial in range(2):
        rng = np.random.default_rng(80 + trial)
        s2 = s + rng.normal(0, 1e-3, s.shape)
        st = world.to_soa(torch.tensor(s2, device="cuda:0"))
        acts = torch.tensor(rng.normal(0, 0.05, (T, a.shape[1], B)), device="cuda:0")
        g = torch.tensor(rng.normal(0, 1, (T + 1, s.shape[1], B)), device="cuda:0")
        gr.state0.copy_(st); gr.actions.copy_(acts); gr.grad_states.copy_(g)
        states, g0, ga = gr.replay()
        torch.cuda.synchronize()
        ref = na.World(md, device="cuda:0")
        states2, sv, status2 = ref.rollout_soa(st, acts, want_saved=True, warm_start=True)
        g02, ga2 = ref.rollout_backward_soa(sv, g)
        assert torch.equal(states, states2) and torch.equal(g0, g02) and torch.equal(ga, ga2) and torch.equal(gr.status, status2)
        assert (status2[0] & 1).all()


def test_rollout_with_two_constrained_groups_equals_chain_of_timesteps_and_oracle():
    """Two balls on the ground = two constrained groups (the MULTI instantiation of the contact kernels) through the rollout entry
    points: warm-started T-step rollout bit-identical to the chain of single steps, first step against the oracle."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import rollout
    from oracle import OracleWorld
    from util import ball_state, ball_world
    md = ball_world("box_first", n_balls=2)
    B, T = 128, 5
    S, A = [], []
    for i in range(B):
        r = np.random.default_rng(900 + i)
        s1, a1 = ball_state(md, [(r.uniform(-1.5, -0.3), r.uniform(-1, 1)), (r.uniform(0.3, 1.5), r.uniform(-1, 1))], 900 + i, pen=float(r.uniform(5e-4, 3e-3)))
        S.append(s1); A.append(a1)
    s0, a0 = np.array(S), np.array(A)
    acts = np.repeat(a0[:, None, :], T, 1) + np.random.default_rng(4).normal(0, 0.05, (B, T, a0.shape[1]))
    w = np.random.default_rng(5).normal(0, 1, (B, T + 1, s0.shape[1]))
    world = na.World(md, device="cuda:0")
    st, at, xs = _chain(world, s0, acts, True)
    (xs * torch.tensor(w, device="cuda:0")).sum().backward()
    world2 = na.World(md, device="cuda:0")
    st2 = torch.tensor(s0, device="cuda:0", requires_grad=True); at2 = torch.tensor(acts, device="cuda:0", requires_grad=True)
    ys = rollout(world2, st2, at2, warm_start=True)
    (ys * torch.tensor(w, device="cuda:0")).sum().backward()
    assert torch.equal(ys, xs.detach())
    for a, b in ((st2.grad, st.grad), (at2.grad, at.grad)):
        assert (a - b).abs().max().item() <= 1e-12 * max(b.abs().max().item(), 1.0)
    ow = OracleWorld(md)
    ref = ow.step_batch(s0, acts[:, 0], None, threads=4)
    assert (ref["status"] & 1).all()
    assert np.abs(ys[:, 1].detach().cpu().numpy() - ref["next"]).max() <= 1e-7 * np.abs(ref["next"]).max()


@pytest.mark.parametrize("contact,warm,K", [(True, True, 4), (True, True, 5), (True, False, 3), (False, True, 4), (True, True, 1)])
def test_checkpointed_rollout_is_bit_identical_and_keeps_only_K_records(contact, warm, K):
    """rollout(checkpoint_every=K): K saved records resident instead of T; the backward pass re-runs the other segments from their
    stored start states and checkpointed LCP warm starts.  The forward kernels are bit-reproducible, so states AND gradients (state0,
    actions, masses) equal the unsegmented rollout bit for bit - with K dividing T or not, with and without warm start, K = 1."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import rollout
    from nimblephysics_amd.world import RolloutRecord
    B, T = 192, 14
    md, s0, a0 = contact_inputs("atlas20", B, 31)
    if not contact:
        md = na.atlas("atlas20", ground=False)
    rng = np.random.default_rng(5)
    acts = np.repeat(a0[:, None, :], T, 1) + rng.normal(0, 0.05, (B, T, a0.shape[1]))
    w = torch.tensor(rng.normal(0, 1, (B, T + 1, s0.shape[1])), device="cuda:0")
    out = []
    for k in (0, K):
        world = na.World(md, device="cuda:0")
        world.tuneMass(1, na.WrtMassBodyNodeEntryType.INERTIA_MASS)
        st = torch.tensor(s0, device="cuda:0", requires_grad=True)
        at = torch.tensor(acts, device="cuda:0", requires_grad=True)
        ms = world.getMasses().clone().requires_grad_(True)
        ys = rollout(world, st, at, warm_start=warm, mass=ms, checkpoint_every=k)
        rec = world.rollout_record
        (ys * w).sum().backward()
        out.append((ys.detach(), st.grad, at.grad, ms.grad, rec, world))
    (y0, gs0, ga0, gm0, rec0, w0), (y1, gs1, ga1, gm1, rec1, w1) = out
    assert torch.equal(y0, y1) and torch.equal(gs0, gs1) and torch.equal(ga0, ga1) and torch.equal(gm0, gm1)
    assert gs0.abs().max().item() > 0 and gm0.abs().max().item() > 0
    # what stays resident: K records (+ the warm starts at the segment boundaries) against T records
    per = w1._L.nbl_saved_bytes(w1._h, B)
    assert isinstance(rec1, RolloutRecord) and rec1.saved.numel() == K * per and rec0.numel() == T * per
    assert rec1.resident_bytes() < (K + 1) * per


@pytest.mark.parametrize("variant,first", [("balls", 310000), ("big", 310100), ("multi", 310200)])
def test_rollout_of_random_mixed_feature_models_equals_the_chain_of_timesteps(variant, first):
    """The T-step driver on the mixed-feature models of the soak (tools/soak_stress.py mix: capsules, enforced joint limits,
    self-collision, frictionless and barely-frictional contacts, action subsets, several skeletons ...): warm-started, with and without
    checkpointing, the states are bit-identical to the chain of single steps and the gradients equal to round-off - joint-limit rows
    and frictionless contacts ride in the carried LCP solution like contacts do."""
    import os
    import sys
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import rollout
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak_parity
    import soak_stress
    B, T, done, limit_rows, nan_worlds = 64, 5, 0, 0, 0
    for seed in range(first, first + 12):
        case = soak_parity.make_case(seed, B, variant == "big", variant == "multi", variant == "balls", False)
        if case is None:
            continue
        md, s0, a0, _ = soak_stress.mutator("mix")(seed, *case)
        try:
            world = na.World(md, device="cuda:0"); world2 = na.World(md, device="cuda:0")
        except na.NimbleAmdError:
            continue
        rng = np.random.default_rng(seed)
        acts = np.repeat(a0[:, None, :], T, 1) + rng.normal(0, 0.05, (B, T, a0.shape[1]))
        w = rng.normal(0, 1, (B, T + 1, s0.shape[1]))
        st, at, xs = _chain(world, s0, acts, True)
        (xs * torch.tensor(w, device="cuda:0")).sum().backward()
        for k in (0, 2):
            st2 = torch.tensor(s0, device="cuda:0", requires_grad=True); at2 = torch.tensor(acts, device="cuda:0", requires_grad=True)
            ys = rollout(world2, st2, at2, warm_start=True, checkpoint_every=k) if k else rollout(world2, st2, at2, warm_start=True)
            (ys * torch.tensor(w, device="cuda:0")).sum().backward()
            # (a world whose state leaves the finite range - a limb driven through its enforced limit at 5 ms steps - is NaN from there on in
            #  both, flagged NBL_ST_NAN, and poisons nothing else: DESIGN.md section 1)
            ok = torch.isfinite(xs.detach()).flatten(1).all(1)
            assert torch.equal(ok, torch.isfinite(ys.detach()).flatten(1).all(1)), (seed, k)
            assert ok.float().mean().item() > 0.9, (seed, ok.float().mean().item())
            assert torch.equal(ys.detach()[ok], xs.detach()[ok]), (seed, k)
            for a, b in ((st2.grad, st.grad), (at2.grad, at.grad)):
                assert torch.isfinite(a[ok]).all() and torch.isfinite(b[ok]).all(), (seed, k)
                assert (a[ok] - b[ok]).abs().max().item() <= 1e-11 * max(b[ok].abs().max().item(), 1.0), (seed, k)
            nan_worlds += int((~ok).sum())
        limit_rows += int((world2.rollout_status.cpu().numpy() & 0x400 != 0).sum())
        done += 1
    print(variant, "models", done, "worlds with joint-limit rows", limit_rows, "non-finite worlds", nan_worlds)
    assert done >= 8
    if variant != "multi":
        assert limit_rows > 0
