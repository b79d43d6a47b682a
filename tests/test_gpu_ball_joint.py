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


def test_mass_gradients_use_the_callers_body_indices():
    """The device model carries two extra massless bodies per ball joint; tuneMass / setMasses / the mass gradient keep the body indices of
    the description (nbl_model's body map): dL/dmass against central differences of the oracle step, bodies before and after ball joints."""
    from nimblephysics_amd.mass import WrtMassBodyNodeEntryType as T
    from test_gpu_mass import _check
    md = ball_model(6, True)
    rng = np.random.default_rng(40)
    n = md.num_dofs
    s = np.concatenate([rng.normal(0, 0.5, (32, n)), rng.normal(0, 1.0, (32, n))], 1); a = rng.normal(0, 1, (32, len(md.action_map)))
    _check(md, [(0, T.INERTIA_MASS), (1, T.INERTIA_FULL), (3, T.INERTIA_COM), (4, T.INERTIA_DIAGONAL)], s, a, 41, tol=5e-6)


def test_rollout_through_ball_joints_equals_the_oracle_and_its_checkpointed_form():
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import rollout
    from oracle import OracleWorld
    md = ball_model(7, True, ground=True)
    B, T = 32, 6
    rng = np.random.default_rng(50)
    n = md.num_dofs; k = len(md.action_map)
    q = rng.normal(0, 0.4, (B, n)); q[:, 3] = 0; q[:, 5] = 0; q[:, 4] = rng.uniform(0.15, 0.4, B)
    s0 = np.concatenate([q, rng.normal(0, 0.3, (B, n))], 1)
    acts = rng.normal(0, 0.5, (B, T, k)); w = rng.normal(0, 1, (B, T + 1, 2 * n))
    outs = []
    for K in (0, 4):
        world = na.World(md, device="cuda:0")
        st = torch.tensor(s0, device="cuda:0", requires_grad=True); at = torch.tensor(acts, device="cuda:0", requires_grad=True)
        ys = rollout(world, st, at, warm_start=False, checkpoint_every=K)
        (ys * torch.tensor(w, device="cuda:0")).sum().backward()
        outs.append((ys.detach().cpu().numpy(), st.grad.cpu().numpy(), at.grad.cpu().numpy()))
    for x, y in zip(outs[0], outs[1]):
        assert np.array_equal(x, y)
    ow = OracleWorld(md)
    worst = 0.0
    for b in range(B):
        worlds = [OracleWorld(md) for _ in range(T)]
        xs = [s0[b]]
        for t in range(T):
            xs.append(worlds[t].step(xs[-1], acts[b, t]))
        g = w[b, T].copy(); ga = np.zeros((T, k))
        for t in reversed(range(T)):
            g, ga[t] = worlds[t].backprop(g)
            g = g + w[b, t]
        sc = lambda r: max(np.abs(r).max(), 1e-30)
        worst = max(worst, np.abs(outs[0][0][b] - np.array(xs)).max() / sc(np.array(xs)), np.abs(outs[0][1][b] - g).max() / sc(g),
                    np.abs(outs[0][2][b] - ga).max() / sc(ga))
    assert worst < 1e-6, worst


def test_c_abi_rejects_ball_models_the_wavefront_kernels_cannot_hold(monkeypatch):
    import nimblephysics_amd as na
    from nimblephysics_amd._lib import NimbleAmdError
    monkeypatch.setenv("NBL_COOP_TREE", "0")
    with pytest.raises(NimbleAmdError):
        na.World(ball_model(8, True), device="cuda:0")


@pytest.mark.parametrize("name", ["serial_chain_ball_joint", "tree_structure_ball_joint"])
def test_the_references_ball_joint_test_worlds_equal_the_oracle(name):
    """data/skel/test/serial_chain_ball_joint.skel (10 ball joints in a row, 30 DOFs) and tree_structure_ball_joint.skel (13 ball joints, 39
    DOFs), transcribed by tools/urdf_to_model.py (tests/test_loaders.py checks the transcription against the reference's files)."""
    import json
    import os
    import nimblephysics_amd as na
    path = os.path.join(os.path.dirname(na.__file__), "data", name + ".json")
    md = na.ModelDescription.from_json(json.load(open(path)))
    assert all(b.joint_type == "ball" for b in md.bodies)
    err, _, _ = _compare(md, 64, 60, vel=2.0)
    assert err.max() < TOL, err.max()


def free_below_root_model(seed, ground=False):
    """A free joint BELOW other bodies (a pallet carried by an arm) and a second one below a free root; frames, inertias and joint
    properties random."""
    import nimblephysics_amd as na
    from test_ball_joint import _T
    rng = np.random.default_rng(400 + seed)
    def body(name, parent, jt, **kw):
        A = rng.normal(size=(3, 3)); I = A @ A.T * 0.02 + 0.03 * np.eye(3)
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        root = parent < 0 and jt == "free"
        return na.BodySpec(name, parent, jt, name + "_joint", axis=tuple(ax) if jt == "revolute" else (0.0, 0.0, 1.0),
                           T_pj=np.eye(4) if root else _T(rng, 0.25), T_cj=np.eye(4) if root else _T(rng, 0.1), mass=float(rng.uniform(0.5, 2.0)),
                           com=tuple(rng.normal(0, 0.04, 3)), inertia=(I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]), **kw)
    bodies = [body("base", -1, "free"), body("arm", 0, "revolute"),
              body("pallet", 1, "free", damping=tuple(rng.uniform(0.1, 1.0, 6)), spring=tuple(rng.uniform(0.5, 3.0, 6)), rest=tuple(rng.normal(0, 0.1, 6))),
              body("box", 2, "revolute"), body("drone", 0, "free")]
    boxes = []
    if ground:
        boxes = [na.BoxSpec(-1, na.make_transform((0, -0.005, 0)), (20.0, 0.01, 20.0), 1.0), na.SphereSpec(2, np.eye(4), 0.12, 0.8),
                 na.BoxSpec(4, np.eye(4), (0.2, 0.15, 0.1), 0.9), na.SphereSpec(0, np.eye(4), 0.1, 0.7)]
    return na.ModelDescription("free_below_root", bodies, boxes, gravity=(0.0, -9.81, 0.0), dt=1e-3, max_contacts=8 if ground else 0)


@pytest.mark.parametrize("vel", [1.0, 12.0])
def test_free_joints_below_the_root_equal_the_oracle(vel):
    """FreeJoint anywhere in the tree (the reference's joint is not tied to the root): on the device six coincident axes (3 rotations,
    3 translations) behind T_pj [exp(q_r), q_p] plus the closed-form term [(wy wz, -wx wz, wx wy); w x u]."""
    md = free_below_root_model(0)
    assert md.num_dofs == 6 + 1 + 6 + 1 + 6
    err, _, _ = _compare(md, 64, 70, vel=vel)
    assert err.max() < TOL, err.max()


def test_free_joints_below_the_root_in_contact_and_through_a_rollout():
    md = free_below_root_model(1, ground=True)
    err, status, rstatus = _compare(md, 256, 71, on_ground=True, vel=0.3)
    assert np.array_equal(status & 1, rstatus & 1) and (status & 1).mean() > 0.2
    ok = ((status | rstatus) & 0x80) == 0
    assert (err[ok] > 1e-5).sum() == 0 and np.median(err[ok]) < TOL, (np.sort(err[ok])[-5:], (err[ok] > TOL).sum())


def _so3_vjp_extended_precision(q, w, dt, g, h=1e-6):
    """posPos^T g and velPos^T g of q' = logMap(exp(q) exp(w dt)) by a five-point stencil in 80-bit arithmetic: logMap itself loses digits
    like 1e-16 / gap^2 next to pi, which a difference quotient in doubles divides by its step."""
    ld = np.longdouble
    def expm(r):
        th = np.sqrt((r * r).sum())
        K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]], dtype=ld)
        if th < 1.0e-3:                                  # the reference's Taylor branch (Geometry.cpp:539-553): part of the function
            return np.eye(3, dtype=ld) + K + ld(0.5) * (K @ K)
        return np.eye(3, dtype=ld) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / (th * th)) * (K @ K)
    def logm(R):
        th = np.arccos((np.trace(R) - 1) / 2)
        return (th / (2 * np.sin(th))) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]], dtype=ld)
    q, w, g = (np.asarray(x, dtype=ld) for x in (q, w, g))
    f = lambda qq, ww: (logm(expm(qq) @ expm(ww * ld(dt))) * g).sum()
    out = np.zeros(6)
    for j in range(3):
        e = np.zeros(3, dtype=ld); e[j] = ld(h)
        out[j] = float((-f(q + 2 * e, w) + 8 * f(q + e, w) - 8 * f(q - e, w) + f(q - 2 * e, w)) / (12 * ld(h)))
        out[3 + j] = float((-f(q, w + 2 * e) + 8 * f(q, w + e) - 8 * f(q, w - e) + f(q, w - 2 * e)) / (12 * ld(h)))
    return out


def test_near_the_log_map_singularity_the_device_is_exact_where_the_references_finite_differences_are_not():
    """The reference finite-differences the position integration of free and ball joints (FreeJoint.cpp:950-1007, BallJoint.cpp:351-408:
    central differences in doubles, eps 1e-6).  Where the next rotation angle comes within ~1e-2 rad of pi, logMap loses digits like
    1e-16 / gap^2 and the quotient divides that by 1e-6: off by 1e-6 .. 1e-4 (measured).  The device carries the exact reverse mode.  With
    a cotangent on the next POSITIONS only, the state gradient of a ball joint's DOFs is exactly posPos^T g / velPos^T g of its
    integration: device and oracle against a five-point stencil in 80-bit arithmetic."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    from test_ball_joint import _ball_offsets
    md = ball_model(10, False, properties=False)
    n = md.num_dofs; B = 32
    rng = np.random.default_rng(90)
    q = rng.normal(0, 0.4, (B, n)); v = rng.normal(0, 0.5, (B, n))
    offs = _ball_offsets(md)
    for o in offs:                                                  # every ball joint 2e-3 .. 2e-2 rad short of pi
        ax = rng.normal(size=(B, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
        q[:, o:o + 3] = ax * (np.pi - rng.uniform(2e-3, 2e-2, (B, 1)))
    s = np.concatenate([q, v], 1); a = np.zeros((B, len(md.action_map)))
    g = np.concatenate([rng.normal(0, 1, (B, n)), np.zeros((B, n))], 1)
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    timestep(world, st, at).backward(torch.tensor(g, device="cuda:0"))
    dev = st.grad.cpu().numpy()
    ref = OracleWorld(md).step_batch(s, a, g, threads=8)["grad_state"]
    e_dev = e_ref = 0.0
    for w in range(B):
        for o in offs:
            x = _so3_vjp_extended_precision(q[w, o:o + 3], v[w, o:o + 3], md.dt, g[w, o:o + 3])
            idx = list(range(o, o + 3)) + list(range(n + o, n + o + 3))
            sc = np.abs(x).max()
            e_dev = max(e_dev, np.abs(dev[w, idx] - x).max() / sc); e_ref = max(e_ref, np.abs(ref[w, idx] - x).max() / sc)
    print(f"vs the extended-precision stencil: device {e_dev:.1e}, oracle (the reference's quotient in doubles) {e_ref:.1e}")
    assert e_dev < 1e-6 and e_ref > 20 * e_dev, (e_dev, e_ref)
