"""Euler / universal / translational / translational-2D / planar joints are expanded into chains of revolute and prismatic joints
through massless virtual links (nimblephysics_amd/model.py).  The expansion is checked INDEPENDENTLY of the chain code:
  * kinematics: every body's world transform equals the product of the reference's closed-form relative transforms
    T_pj * M(q) * T_cj^-1 (EulerJoint.cpp:1333 + Geometry.cpp:1767-1797, UniversalJoint.cpp:193, TranslationalJoint.cpp:127,
    TranslationalJoint2D.cpp:232, PlanarJoint.cpp:296), written out here with numpy / scipy;
  * inertia: the mass matrix equals sum_b J_b^T G_b J_b with the body Jacobians J_b taken by finite differences of those
    closed-form transforms (so the massless links contribute nothing and the real bodies exactly what they should);
  * the reference's property tests (equations of motion, Featherstone Jacobians vs FD) on the expanded model."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import nimblephysics_amd as na
from nimblephysics_amd.model import BodySpec, ModelDescription, _inv, _inertia_matrix
from oracle import OracleWorld
from util import rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
SKEL = os.path.join(HERE, "golden", "compound_joints.skel")


def _T(R=None, p=None):
    T = np.eye(4)
    if R is not None: T[:3, :3] = R
    if p is not None: T[:3, 3] = p
    return T


def _rot(axis, a):
    axis = np.asarray(axis, float); axis = axis / np.linalg.norm(axis)
    return Rotation.from_rotvec(axis * a).as_matrix()


def closed_form_motion(b: BodySpec, q):
    """M(q) of the unexpanded joint b, straight from the reference's updateRelativeTransform of each joint class."""
    jt = b.joint_type
    if jt == "revolute": return _T(_rot(b.axis, q[0]))
    if jt == "prismatic": return _T(p=np.asarray(b.axis, float) / np.linalg.norm(b.axis) * q[0])
    if jt.startswith("euler_"):
        return _T(Rotation.from_euler(jt[6:].upper(), q).as_matrix())     # intrinsic: R = R_a(q0) R_b(q1) R_c(q2)
    if jt == "universal": return _T(_rot(b.axes[0], q[0]) @ _rot(b.axes[1], q[1]))
    if jt == "translational": return _T(p=np.asarray(q, float))
    a1 = np.asarray(b.axes[0], float) / np.linalg.norm(b.axes[0]); a2 = np.asarray(b.axes[1], float) / np.linalg.norm(b.axes[1])
    if jt == "translational2d": return _T(p=a1 * q[0] + a2 * q[1])
    if jt == "planar":
        d = a1 @ a2
        if abs(d) > 1e-6: a2 = a2 - d * a1; a2 /= np.linalg.norm(a2)
        rot = np.cross(a1, a2); rot /= np.linalg.norm(rot)
        return _T(p=a1 * q[0] + a2 * q[1]) @ _T(_rot(rot, q[2]))
    raise AssertionError(jt)


NDOF = {"revolute": 1, "prismatic": 1, "universal": 2, "translational": 3, "translational2d": 2, "planar": 3}


def closed_form_fk(bodies, q):
    out, off = [], 0
    for b in bodies:
        k = 3 if b.joint_type.startswith("euler_") else NDOF[b.joint_type]
        Tp = np.eye(4) if b.parent < 0 else out[b.parent]
        out.append(Tp @ np.asarray(b.T_pj) @ closed_form_motion(b, q[off:off + k]) @ _inv(np.asarray(b.T_cj)))
        off += k
    return out


def _skel_bodies():
    """The fixture twice: as the loader emits it (compound joints intact) and expanded."""
    from nimblephysics_amd import loaders
    captured = {}
    orig = loaders.ModelDescription

    def spy(name, bodies, boxes, *a, **kw):
        captured["bodies"] = [BodySpec(**{**b.__dict__}) for b in bodies]
        return orig(name, bodies, boxes, *a, **kw)
    loaders.ModelDescription = spy
    try:
        md = na.load_skel(SKEL)
    finally:
        loaders.ModelDescription = orig
    return captured["bodies"], md


def test_skel_compound_joints_expand_to_the_reference_kinematics():
    raw, md = _skel_bodies()
    assert [b.joint_type for b in raw] == ["planar", "euler_xyz", "universal", "translational2d", "euler_zyx", "translational"]
    assert md.num_dofs == 3 + 3 + 2 + 2 + 3 + 3 and len(md.bodies) == md.num_dofs       # one 1-DOF joint per coordinate
    assert sum(b.mass == 0.0 for b in md.bodies) == md.num_dofs - len(raw)                 # the rest are massless virtual links
    fl = md.flat()
    assert fl["damping"][:3].tolist() == [0.1, 0.0, 0.05] and fl["spring"][2] == 2.0 and fl["rest"][2] == 0.1
    assert fl["pos_lo"][3] == -1.0 and fl["pos_hi"][3] == 1.5 and fl["damping"][3:6].tolist() == [0.2, 0.3, 0.0]
    w = OracleWorld(md)
    rng = np.random.default_rng(0)
    for _ in range(5):
        q = rng.normal(0, 0.7, md.num_dofs)
        ref = closed_form_fk(raw, q)
        for i, T in enumerate(ref):
            assert np.abs(w.body_world_transform(q, md.body_index[i]) - T).max() < 1e-13, (i, raw[i].joint_type)


def test_mass_matrix_equals_the_closed_form_kinematics_route():
    raw, md = _skel_bodies()
    w = OracleWorld(md)
    n = md.num_dofs
    q = np.random.default_rng(1).normal(0, 0.6, n)

    def body_jac(i):       # body-frame spatial Jacobian of raw body i by central differences of the closed-form FK
        T0 = closed_form_fk(raw, q)[i]
        J = np.zeros((6, n)); eps = 1e-6
        for d in range(n):
            qp, qm = q.copy(), q.copy(); qp[d] += eps; qm[d] -= eps
            dT = (closed_form_fk(raw, qp)[i] - closed_form_fk(raw, qm)[i]) / (2 * eps)
            X = _inv(T0) @ dT                                  # se(3) element [w]x, v in the body frame
            J[:, d] = [X[2, 1], X[0, 2], X[1, 0], X[0, 3], X[1, 3], X[2, 3]]
        return J
    M = np.zeros((n, n))
    for i, b in enumerate(raw):
        c = np.asarray(b.com, float)
        cx = np.array([[0, -c[2], c[1]], [c[2], 0, -c[0]], [-c[1], c[0], 0]])
        G = np.zeros((6, 6))
        G[:3, :3] = _inertia_matrix(b.inertia) + b.mass * cx @ cx.T
        G[:3, 3:] = b.mass * cx; G[3:, :3] = b.mass * cx.T; G[3:, 3:] = b.mass * np.eye(3)
        J = body_jac(i)
        M += J.T @ G @ J
    assert rel_err(w.mass_matrix(q), M) < 1e-7


def test_property_tests_on_the_expanded_model():
    _, md = _skel_bodies()
    w = OracleWorld(md); n = w.n
    rng = np.random.default_rng(2)
    fl = md.flat()
    for _ in range(2):
        q, v, tau = rng.normal(0, 0.5, n), rng.normal(0, 0.5, n), rng.normal(0, 1, n)
        M = w.mass_matrix(q)
        assert np.abs(M - M.T).max() < 1e-12 and np.linalg.eigvalsh(M).min() > 0
        qdd = w.forward_dynamics(q, v, tau)
        rhs = tau - w.coriolis_gravity(q, v) - fl["damping"] * v - fl["spring"] * (q - fl["rest"] + md.dt * v)
        assert rel_err(M @ qdd, rhs) < 1e-9
    # VJP == J^T g through a step
    s = np.concatenate([rng.normal(0, 0.4, n), rng.normal(0, 0.4, n)])[None]; a = rng.normal(0, 1, (1, n)); g = rng.normal(0, 1, (1, 2 * n))
    r = w.step_batch(s, a, g)
    eps = 1e-6
    fd = np.zeros(2 * n)
    for d in range(2 * n):
        sp, sm = s.copy(), s.copy(); sp[0, d] += eps; sm[0, d] -= eps
        fd[d] = ((w.step_batch(sp, a)["next"] - w.step_batch(sm, a)["next"]) / (2 * eps) * g).sum()
    assert rel_err(r["grad_state"][0], fd) < 1e-6


@pytest.mark.skipif(not os.path.isdir("/root/reference/data/skel/test"), reason="reference data not present")
@pytest.mark.parametrize("name,ndof", [("test/single_pendulum_euler_joint.skel", 3), ("test/double_pendulum_euler_joint.skel", 6),
                                        ("test/serial_chain_eulerxyz_joint.skel", None), ("test/planar_joint.skel", None),
                                        ("test/translational_joints.skel", None)])
def test_the_references_own_compound_joint_skel_files_load(name, ndof):
    md = na.load_skel(os.path.join("/root/reference/data/skel", name))
    assert md.num_dofs == len([b for b in md.bodies if b.joint_type in ("revolute", "prismatic")]) + 6 * sum(b.joint_type == "free" for b in md.bodies)
    if ndof is not None:
        assert md.num_dofs == ndof
    w = OracleWorld(md)
    q = np.random.default_rng(3).normal(0, 0.3, md.num_dofs)
    M = w.mass_matrix(q)
    assert np.abs(M - M.T).max() < 1e-10 and np.linalg.eigvalsh(M).min() > 0
