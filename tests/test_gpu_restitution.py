"""Restitution / bounce (ContactConstraint.cpp:95-110, 395-442; bounce diagonals CGGM.cpp:770; bounce approximation of the
position Jacobians BackpropSnapshot.cpp:1131-1226) on the device against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _fwd_bwd(md, s, a, seed):
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
    g = np.random.default_rng(seed).normal(0, 1, s.shape)
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy().astype(np.uint32)
    out.backward(torch.tensor(g, device="cuda:0"))
    ref = ow.step_batch(s, a, g, threads=8)
    dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
    errs = {k: np.abs(dev[k] - ref[k]).max(1) / max(np.abs(ref[k]).max(), 1e-30) for k in dev}
    return dev, ref, errs, status, world


def _balls(B, seed, n_balls, e_ground, e_ball, speed):
    from util import ball_state, ball_world
    md = ball_world("box_first", n_balls=n_balls)
    md.boxes[0].restitution = e_ground
    for bx in md.boxes[1:]:
        bx.restitution = e_ball
    n = md.num_dofs
    S, A = [], []
    for i in range(B):
        r = np.random.default_rng(seed * 1000 + i)
        centres = [(r.uniform(-1.5, -0.3), r.uniform(-1, 1)), (r.uniform(0.3, 1.5), r.uniform(-1, 1))][:n_balls]
        s1, a1 = ball_state(md, centres, seed * 1000 + i, pen=float(r.uniform(5e-4, 3e-3)))
        for k in range(n_balls):
            s1[6 * k:6 * k + 3] = r.normal(0, 0.3, 3)
            # world-frame downward velocity `speed` expressed in the rotated body frame of the free joint
            from scipy.spatial.transform import Rotation
            R = Rotation.from_rotvec(s1[6 * k:6 * k + 3]).as_matrix()
            s1[n + 6 * k + 3:n + 6 * k + 6] = R.T @ np.array([r.normal(0, 0.05), -speed * r.uniform(0.8, 1.2), r.normal(0, 0.05)])
        S.append(s1); A.append(a1)
    return md, np.array(S), np.array(A)


@pytest.mark.parametrize("n_balls,e_ground,e_ball,speed", [(1, 0.9, 0.8, 1.0), (2, 1.0, 0.5, 2.0), (1, 0.9, 0.8, 0.05)])
def test_bouncing_balls_fwd_bwd_vs_oracle(n_balls, e_ground, e_ball, speed):
    """Balls hitting the ground with e = e_A e_B: above the 0.1 m/s bounce threshold the normal row gets b += e b (the ball leaves with
    ~e times its approach speed), the backward pass carries the bounce diagonals 1 + e and the bounce approximation of posPos /
    velPos; below the threshold (last case: e b = 0.036 < 0.1) nothing bounces and the step is the plain inelastic one."""
    md, s, a = _balls(128, 5, n_balls, e_ground, e_ball, speed)
    dev, ref, errs, status, world = _fwd_bwd(md, s, a, 6)
    assert (status & 0x1).all()
    e = e_ground * e_ball
    n = md.num_dofs
    from scipy.spatial.transform import Rotation
    # world-frame vertical velocity of ball 0 after the step
    vy_after = np.array([(Rotation.from_rotvec(dev["next"][i, 0:3]).as_matrix() @ dev["next"][i, n + 3:n + 6])[1] for i in range(len(s))])
    vy_before = np.array([(Rotation.from_rotvec(s[i, 0:3]).as_matrix() @ s[i, n + 3:n + 6])[1] for i in range(len(s))])
    if e * speed * 0.8 > 0.1:
        assert np.all(vy_after > 0.5 * e * np.abs(vy_before))              # it really bounced
    else:
        assert np.all(np.abs(vy_after) < 1e-4)                             # inelastic
    for k, v in errs.items():
        assert v.max() < TOL, (k, v.max())


def test_bouncing_cube_on_four_corners():
    """A cube dropped flat on the ground: four coplanar bouncing contacts (a rank-deficient Gram matrix of the bounce approximation
    is handled by the pseudo-inverse); friction and bounce together."""
    from util import box_stack_inputs
    md, s, a = box_stack_inputs(256, 91, overhang=False)
    for bx in md.boxes:
        bx.restitution = 0.8
    n = md.num_dofs
    s[:, 10] += 5.0                      # park the second cube far above
    s[:, n + 4] = -np.random.default_rng(3).uniform(0.5, 1.5, len(s))   # cube 1 falls at 0.5 .. 1.5 m/s (yaw-only rotation: y is y)
    dev, ref, errs, status, world = _fwd_bwd(md, s, a, 92)
    assert (status & 0x1).all()
    assert np.all(dev["next"][:, n + 4] > 0.25)          # 0.64 x the approach speed minus what the LCP distributes
    print("[bouncing cube] max errs", {k: float(v.max()) for k, v in errs.items()}, "worlds above 1e-5:", {k: int((v > 1e-5).sum()) for k, v in errs.items()})
    assert errs["next"].max() < TOL
    assert (errs["grad_state"] > 1e-5).mean() <= 0.02 and (errs["grad_action"] > 1e-5).mean() <= 0.02
    assert np.median(errs["grad_state"]) < 1e-8


def test_restitution_off_is_bitwise_the_default_path():
    """Coefficients whose product stays below 1e-3 (DART_RESTITUTION_COEFF_THRESHOLD) change nothing, bit for bit."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    md, s, a = _balls(64, 7, 1, 0.0, 0.0, 1.0)
    md2, _, _ = _balls(64, 7, 1, 0.03, 0.03, 1.0)
    outs = []
    for m in (md, md2):
        w = na.World(m, device="cuda:0")
        outs.append(timestep(w, torch.tensor(s, device="cuda:0"), torch.tensor(a, device="cuda:0")))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("e_ball", [0.0, 0.8])
def test_penetration_correction_fwd_bwd_vs_oracle(e_ball):
    """World::setPenetrationCorrectionEnabled(true) (ContactConstraint.cpp:393-441): the normal row gets min(depth * 0.01 / dt, 1e-3)
    unless the restitution velocity is larger; the backward pass treats it as a constant, like the reference.  Slow balls (below the
    bounce threshold: the correction applies) and fast ones (restitution wins) against the oracle, and the World-mirror switch."""
    import torch
    import nimblephysics_amd as na
    md, s, a = _balls(128, 9, 1, 0.9, e_ball, 0.05)
    md2, s2, a2 = _balls(128, 10, 1, 0.9, e_ball, 1.5)
    s = np.concatenate([s, s2]); a = np.concatenate([a, a2])
    off, _, _, _, world = _fwd_bwd(md, s, a, 6)
    assert not world.getPenetrationCorrectionEnabled()
    md.penetration_correction = True
    dev, ref, errs, status, world = _fwd_bwd(md, s, a, 6)
    assert world.getPenetrationCorrectionEnabled() and (status & 0x1).all()
    for k, v in errs.items():
        assert v.max() < TOL, (k, v.max())
    # the correction changed the step of the slow balls (depth 0.5-3 mm -> capped at 1e-3 m/s of extra separation speed) ...
    d = np.abs(dev["next"] - off["next"]).max(1)
    assert d[:128].min() > 1e-5 and d[:128].max() < 5e-2
    # ... and where the ball bounces harder than the correction, restitution overrides it (the step is unchanged)
    if e_ball > 0:
        assert d[128:].max() == 0.0
    # the switch on the World mirror re-uploads the constants
    world.setPenetrationCorrectionEnabled(False)
    st = torch.tensor(s, device="cuda:0"); at = torch.tensor(a, device="cuda:0")
    from nimblephysics_amd.timestep import timestep
    assert np.array_equal(timestep(world, st, at).cpu().numpy(), off["next"])
