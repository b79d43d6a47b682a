"""Ball joints (dart/dynamics/BallJoint.cpp, DART_USE_IDENTITY_JACOBIAN build: exponential-map positions, child-frame angular velocity,
S = Ad(T_cj)[:, 0:3], q' = log(exp(q) exp(v dt))) in the CPU oracle, pinned the way the reference pins its joints
(unittests/GradientTestUtils.hpp: every analytical quantity against finite differences of the same engine), plus the identity the device
library builds on: at the velocity level a ball joint IS three coincident single-axis joints x, y, z at zero angle behind T_pj exp(q)."""
import copy

import numpy as np
import pytest

import nimblephysics_amd as na
from oracle import OracleWorld
from test_oracle_props import _fd_jac
from util import rel_err


def _T(rng, s):
    from scipy.spatial.transform import Rotation as Rot
    T = np.eye(4); T[:3, :3] = Rot.from_rotvec(rng.normal(0, 0.6, 3)).as_matrix(); T[:3, 3] = rng.normal(0, s, 3)
    return T


def ball_model(seed=0, free_root=True, ground=False, properties=True):
    """free (or revolute) root -> ball -> revolute -> ball, a second branch with a ball on the root; random frames, inertias and (optionally)
    damping / springs on the ball DOFs; with `ground`: sphere and box colliders on the links above a ground box."""
    rng = np.random.default_rng(100 + seed)
    def body(name, parent, jt, **kw):
        A = rng.normal(size=(3, 3)); I = A @ A.T * 0.02 + 0.03 * np.eye(3)
        nd = {"free": 6, "ball": 3}.get(jt, 1)
        extra = {}
        if properties and jt == "ball":
            extra = dict(damping=tuple(rng.uniform(0.1, 1.0, nd)), spring=tuple(rng.uniform(0.5, 3.0, nd)), rest=tuple(rng.normal(0, 0.1, nd)))
        root = parent < 0 and jt == "free"
        ax = tuple(np.eye(3)[int(rng.integers(0, 3))])
        return na.BodySpec(name, parent, jt, name + "_joint", axis=ax if jt == "revolute" else (0.0, 0.0, 1.0),
                           T_pj=np.eye(4) if root else _T(rng, 0.25), T_cj=np.eye(4) if root else _T(rng, 0.1),
                           mass=float(rng.uniform(0.5, 2.0)), com=tuple(rng.normal(0, 0.04, 3)),
                           inertia=(I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]), **extra, **kw)
    bodies = [body("root", -1, "free" if free_root else "revolute"), body("upper", 0, "ball"), body("fore", 1, "revolute"),
              body("hand", 2, "ball"), body("tail", 0, "ball")]
    boxes = []
    if ground:
        boxes = [na.BoxSpec(-1, na.make_transform((0, -0.005, 0)), (20.0, 0.01, 20.0), 1.0),
                 na.SphereSpec(3, na.make_transform((0.02, 0, 0)), 0.12, 0.8), na.SphereSpec(4, np.eye(4), 0.1, 0.6),
                 na.BoxSpec(1, np.eye(4), (0.2, 0.15, 0.1), 0.9)]
    return na.ModelDescription("ball_arm", bodies, boxes, gravity=(0.0, -9.81, 0.0), dt=1e-3, max_contacts=8 if ground else 0)


def _state(md, rng, on_ground=False):
    n = md.num_dofs
    q = rng.normal(0, 0.5, n); v = rng.normal(0, 1.0, n)
    if on_ground:
        q[3:6] = [0.0, 0.18, 0.0]
    return q, v, rng.normal(0, 1.0, len(md.action_map))


def test_ball_joint_kinematics_and_integration_follow_the_exponential_map():
    from scipy.spatial.transform import Rotation as Rot
    md = ball_model(1, free_root=False, properties=False)
    w = OracleWorld(md); rng = np.random.default_rng(2)
    q, v, _ = _state(md, rng)
    assert md.num_dofs == 1 + 3 + 1 + 3 + 3
    qn = w.integrate_positions(q, v)
    for o in (1, 5, 8):                                 # the three ball joints: R' = R(q) R(v dt)  (BallJoint.cpp:333-349)
        Rn = Rot.from_rotvec(q[o:o + 3]).as_matrix() @ Rot.from_rotvec(v[o:o + 3] * md.dt).as_matrix()
        assert np.allclose(Rot.from_rotvec(qn[o:o + 3]).as_matrix(), Rn, atol=1e-13)
    assert np.isclose(qn[0], q[0] + md.dt * v[0]) and np.isclose(qn[4], q[4] + md.dt * v[4])


@pytest.mark.parametrize("free_root", [True, False])
def test_equations_of_motion_and_featherstone_jacobians_with_ball_joints(free_root):
    md = ball_model(2, free_root)
    w = OracleWorld(md); n = w.n; rng = np.random.default_rng(3)
    q, v, tau_a = _state(md, rng)
    tau = np.zeros(n); tau[list(md.action_map)] = tau_a
    fl = md.flat()
    M = w.mass_matrix(q)
    assert np.abs(M - M.T).max() < 1e-12 and np.linalg.eigvalsh(M).min() > 0
    qdd = w.forward_dynamics(q, v, tau)
    rhs = tau - w.coriolis_gravity(q, v) - fl["damping"] * v - fl["spring"] * (q - fl["rest"] + md.dt * v)
    assert rel_err(M @ qdd, rhs) < 1e-10
    scale = max(1.0, np.abs(w.coriolis_gravity(q, v)).max())
    assert np.abs(w.jac_C(q, v, 0) - _fd_jac(lambda x: w.coriolis_gravity(x, v), q)).max() < 2e-7 * scale
    assert np.abs(w.jac_C(q, v, 1) - _fd_jac(lambda x: w.coriolis_gravity(q, x), v)).max() < 2e-7 * scale
    x = rng.normal(0, 1, n)
    assert np.abs(w.jac_Mx(q, x) - _fd_jac(lambda y: w.mass_matrix(y) @ x, q)).max() < 2e-7 * scale


@pytest.mark.parametrize("ground", [False, True])
def test_backprop_through_ball_joints_is_the_jacobian_transpose(ground):
    md = ball_model(3, True, ground)
    w = OracleWorld(md); n = w.n; rng = np.random.default_rng(4)
    q, v, a0 = _state(md, rng, on_ground=ground)
    if ground:
        v *= 0.2
    s0 = np.concatenate([q, v])
    nxt = w.step(s0, a0)
    if ground:
        assert w.last_status & 1                        # in contact
    Js = _fd_jac(lambda x: w.step(x, a0), s0, 1e-6)
    Ja = _fd_jac(lambda x: w.step(s0, x), a0, 1e-6)
    tol = 2e-5 if ground else 1e-6                      # the contact Jacobians carry the reference's own approximations
    for g in (rng.normal(0, 1, 2 * n), np.ones(2 * n)):
        w.step(s0, a0)
        gs, ga = w.backprop(g)
        assert np.abs(gs - Js.T @ g).max() < tol * max(1.0, np.abs(gs).max())
        assert np.abs(ga - Ja.T @ g).max() < tol * max(1.0, np.abs(ga).max())


def chain_of(md, q):
    """The model the device library runs for `md` AT configuration q: every ball joint replaced by three revolute joints x, y, z at zero
    angle (massless first two links) behind T_pj exp(q_ball).  Same DOF numbering."""
    from scipy.spatial.transform import Rotation as Rot
    bodies, new_index, o = [], {}, 0
    for i, b in enumerate(md.bodies):
        nd = md.joint_ndof(i)
        par = -1 if b.parent < 0 else new_index[b.parent]
        if b.joint_type != "ball":
            nb = copy.deepcopy(b); nb.parent = par
            bodies.append(nb)
        else:
            R = np.eye(4); R[:3, :3] = Rot.from_rotvec(q[o:o + 3]).as_matrix()
            for k in range(3):
                last = k == 2
                bodies.append(na.BodySpec(b.name if last else f"{b.name}#{k}", par, "revolute", f"{b.joint_name}#{k}", axis=tuple(np.eye(3)[k]),
                                          T_pj=np.array(b.T_pj) @ R if k == 0 else np.eye(4), T_cj=np.array(b.T_cj) if last else np.eye(4),
                                          mass=b.mass if last else 0.0, com=b.com if last else (0, 0, 0), inertia=b.inertia if last else (0,) * 6,
                                          damping=(b.damping[k],) if b.damping else (), spring=(), rest=()))
                par = len(bodies) - 1
        new_index[i] = len(bodies) - 1
        o += nd
    return na.ModelDescription(md.name + "_chain", bodies, [], gravity=md.gravity, dt=md.dt, max_contacts=0)


def _ball_offsets(md):
    out, o = [], 0
    for i, b in enumerate(md.bodies):
        if b.joint_type == "ball":
            out.append(o)
        o += md.joint_ndof(i)
    return out


def test_a_ball_joint_is_three_coincident_single_axis_joints_plus_a_closed_form_acceleration_term():
    """What the device library builds on.  Mass matrix (hence impulse tests, M^-1, contact Jacobians) of the ball model equal those of the
    x-y-z chain at zero angle behind T_pj exp(q) with the ball's angular velocity as joint rates.  The chain's axes turn with its own
    rates, the ball's do not: the child accelerates equally in both when  qdd_ball = qdd_chain + (wy wz, -wx wz, wx wy),  the Lie
    brackets of the chain's own axis velocities - a term that depends on the ball's velocity only."""
    md = ball_model(5, True, properties=False)
    rng = np.random.default_rng(6)
    q, v, _ = _state(md, rng)
    w = OracleWorld(md); n = w.n
    ch = chain_of(md, q); wc = OracleWorld(ch)
    qc = q.copy(); delta = np.zeros(n)
    for o in _ball_offsets(md):
        qc[o:o + 3] = 0.0
        wx, wy, wz = v[o:o + 3]
        delta[o:o + 3] = [wy * wz, -wx * wz, wx * wy]
    tau = rng.normal(0, 1, n)
    M = w.mass_matrix(q)
    assert rel_err(wc.mass_matrix(qc), M) < 1e-12
    assert rel_err(wc.forward_dynamics(qc, v, tau) + delta, w.forward_dynamics(q, v, tau)) < 1e-10
    assert rel_err(wc.coriolis_gravity(qc, v) - M @ delta, w.coriolis_gravity(q, v)) < 1e-10
