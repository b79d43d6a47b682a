"""Screw joints (dart/dynamics/ScrewJoint.cpp: one DOF, rotation about an axis coupled with `pitch` of translation per turn along it,
S = Ad(T_cj)[axis; axis pitch / 2 pi], T = T_pj expMap(S_local q) T_cj^-1): the oracle against the closed-form transform and finite
differences of its own step (CPU), the device against the oracle (GPU)."""
import numpy as np
import pytest

import nimblephysics_amd as na
from oracle import OracleWorld
from test_oracle_props import _fd_jac
from util import rel_err


def screw_arm(seed=0, free_root=True):
    from test_ball_joint import _T
    rng = np.random.default_rng(300 + seed)
    def body(name, parent, jt, **kw):
        A = rng.normal(size=(3, 3)); I = A @ A.T * 0.02 + 0.03 * np.eye(3)
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        root = parent < 0 and jt == "free"
        return na.BodySpec(name, parent, jt, name + "_joint", axis=tuple(ax) if jt in ("revolute", "screw") else (0.0, 0.0, 1.0),
                           T_pj=np.eye(4) if root else _T(rng, 0.25), T_cj=np.eye(4) if root else _T(rng, 0.1), mass=float(rng.uniform(0.5, 2.0)),
                           com=tuple(rng.normal(0, 0.04, 3)), inertia=(I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]), **kw)
    bodies = [body("root", -1, "free" if free_root else "screw", **({} if free_root else {"pitch": 0.3})),
              body("bolt", 0, "screw", pitch=0.25, damping=(0.2,)), body("link", 1, "revolute"), body("nut", 2, "screw", pitch=-1.5, spring=(2.0,), rest=(0.1,)),
              body("dflt", 0, "screw")]
    return na.ModelDescription("screw_arm", bodies, [], gravity=(0.0, -9.81, 0.0), dt=1e-3, max_contacts=0)


def test_screw_joint_transform_is_a_rotation_with_its_share_of_the_pitch():
    from scipy.spatial.transform import Rotation as Rot
    md = screw_arm(0, free_root=False)
    w = OracleWorld(md); rng = np.random.default_rng(1)
    q = rng.normal(0, 1.0, md.num_dofs)
    b0 = md.bodies[0]
    Q = np.eye(4); ax = np.array(b0.axis); Q[:3, :3] = Rot.from_rotvec(ax * q[0]).as_matrix(); Q[:3, 3] = ax * b0.pitch * q[0] / (2 * np.pi)
    T = np.array(b0.T_pj) @ Q @ np.linalg.inv(np.array(b0.T_cj))
    assert np.allclose(w.body_world_transform(q, 0), T, atol=1e-13)
    assert md.bodies[4].pitch == 0.1                                    # ScrewJointAspect's default


@pytest.mark.parametrize("free_root", [True, False])
def test_screw_joint_dynamics_and_backprop_vs_finite_differences(free_root):
    md = screw_arm(1, free_root)
    w = OracleWorld(md); n = w.n; rng = np.random.default_rng(2)
    q, v, a0 = rng.normal(0, 0.6, n), rng.normal(0, 1.0, n), rng.normal(0, 1.0, len(md.action_map))
    tau = np.zeros(n); tau[list(md.action_map)] = a0
    fl = md.flat()
    M = w.mass_matrix(q)
    rhs = tau - w.coriolis_gravity(q, v) - fl["damping"] * v - fl["spring"] * (q - fl["rest"] + md.dt * v)
    assert rel_err(M @ w.forward_dynamics(q, v, tau), rhs) < 1e-10
    scale = max(1.0, np.abs(w.coriolis_gravity(q, v)).max())
    assert np.abs(w.jac_C(q, v, 0) - _fd_jac(lambda x: w.coriolis_gravity(x, v), q)).max() < 2e-7 * scale
    assert np.abs(w.jac_Mx(q, v) - _fd_jac(lambda y: w.mass_matrix(y) @ v, q)).max() < 2e-7 * scale
    s0 = np.concatenate([q, v])
    Js = _fd_jac(lambda x: w.step(x, a0), s0, 1e-6); Ja = _fd_jac(lambda x: w.step(s0, x), a0, 1e-6)
    g = rng.normal(0, 1, 2 * n)
    w.step(s0, a0)
    gs, ga = w.backprop(g)
    assert np.abs(gs - Js.T @ g).max() < 1e-6 * max(1.0, np.abs(gs).max()) and np.abs(ga - Ja.T @ g).max() < 1e-6 * max(1.0, np.abs(ga).max())


@pytest.mark.gpu
@pytest.mark.parametrize("free_root", [True, False])
def test_screw_joints_on_the_device_equal_the_oracle(free_root):
    import torch
    from nimblephysics_amd.timestep import timestep
    md = screw_arm(2, free_root)
    B = 64; rng = np.random.default_rng(3); n = md.num_dofs
    s = np.concatenate([rng.normal(0, 0.6, (B, n)), rng.normal(0, 1.5, (B, n))], 1); a = rng.normal(0, 1, (B, len(md.action_map))); g = rng.normal(0, 1, s.shape)
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at); out.backward(torch.tensor(g, device="cuda:0"))
    ref = OracleWorld(md).step_batch(s, a, g, threads=8)
    sc = lambda x: max(np.abs(x).max(), 1e-30)
    assert np.abs(out.detach().cpu().numpy() - ref["next"]).max() / sc(ref["next"]) < 1e-7
    assert np.abs(st.grad.cpu().numpy() - ref["grad_state"]).max() / sc(ref["grad_state"]) < 1e-7
    assert np.abs(at.grad.cpu().numpy() - ref["grad_action"]).max() / sc(ref["grad_action"]) < 1e-7
