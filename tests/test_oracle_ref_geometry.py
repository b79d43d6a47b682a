"""The dynamics half of the oracle against REFERENCE CODE: the reference's own spatial-algebra
primitives (dart/math/Geometry.cpp: expMapRot :539, expMapJac :556, logMap :720, AdT :1300, AdInvT :1437, AdInvRLinear :1461, ad :1469,
dAdT :1504, dAdInvT :1530, dad :3506, transformInertia :3515, expMap :3357, expAngular :3414, makeSkewSymmetric :3860) and its flat-array
articulated-body algorithm (dart/dynamics/SimpleFeatherstone.cpp:26-138) are compiled from the reference's files where they lie into
oracle/_ref/libgeometry_ref.so (oracle/ref_build.py: build_geometry; a small fixed-size matrix class stands in for Eigen, which is not
on this machine) and run next to oracle/spatial.hpp / oracle/dynamics.hpp.

Element-wise code must agree BIT FOR BIT (the Taylor branches, expAngular / expMap, logMap including its branch next to pi, the cross
products); code that goes through 3-term inner products to a few ulps (Eigen may associate those sums differently from the stand-in,
see oracle/ref_geometry_prelude.hpp); the oracle's ABA - a different program: Atlas-style tree descriptions, welds, body-frame
recursion with gravity and damping terms - against the reference's flat-array ABA on random trees of one-DOF joints to 1e-14."""
import ctypes as C
import os

import numpy as np
import pytest

import nimblephysics_amd as na
import oracle
from oracle import OracleWorld

REF = os.path.join(os.path.dirname(oracle.__file__), "_ref", "libgeometry_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libgeometry_ref.so not built (python oracle/ref_build.py, needs /root/reference)")

PD = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(PD)


@pytest.fixture(scope="module")
def ref():
    return C.CDLL(REF)


def _oracle_prim(name, a, b, nout):
    lib = oracle._lib()
    lib.nbo_prim.argtypes = [C.c_char_p, PD, PD, PD]
    lib.nbo_prim.restype = C.c_int
    out = np.zeros(nout)
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b if b is not None else np.zeros(1), dtype=np.float64)
    assert lib.nbo_prim(name.encode(), _p(a), _p(b), _p(out)) == nout, name
    return out


def _ref_call(ref, name, a, b, nout):
    out = np.zeros(nout)
    a = np.ascontiguousarray(a, dtype=np.float64)
    fn = getattr(ref, "ref_" + name)
    fn.restype = None
    if b is None:
        fn.argtypes = [PD, PD]; fn(_p(a), _p(out))
    else:
        b = np.ascontiguousarray(b, dtype=np.float64)
        fn.argtypes = [PD, PD, PD]; fn(_p(a), _p(b), _p(out))
    return out


def _ulps(x, y):
    """max |x - y| in units of the last place of the larger magnitude of the pair's vector (0 = bit-identical)."""
    scale = np.maximum(np.abs(x), np.abs(y)).max()
    return 0.0 if np.array_equal(x, y) else float(np.abs(x - y).max() / (np.spacing(scale) if scale > 0 else 5e-324))


def _rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _t12(rng, scale=1.0):
    return np.concatenate([_rot(rng).reshape(9), rng.normal(0, scale, 3)])


def test_exponential_and_log_maps_are_bit_identical_to_the_references(ref):
    rng = np.random.default_rng(1)
    qs = [rng.normal(0, s, 3) for s in (1.0, 3.0, 0.3) for _ in range(200)]
    # the Taylor branches of expMapRot / expMapJac switch at |q| = 1e-3, expAngular's at 1e-6: both sides of each, and zero
    for mag in (0.0, 1e-12, 9.99e-7, 1.001e-6, 5e-4, 9.999e-4, 1.0001e-3, 2e-3):
        for _ in range(20):
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            qs.append(d * mag)
    worst = {"expMapRot": 0.0, "expMapJac": 0.0}
    for q in qs:
        for f in ("expAngular", "makeSkewSymmetric"):
            got = _oracle_prim(f, q, None, 9)
            want = _ref_call(ref, f, q, None, 9 if f == "makeSkewSymmetric" else 12)[:9]
            assert np.array_equal(got, want), (f, q)
        for f in ("expMapRot", "expMapJac"):     # I + a [q] + b [q]^2 with [q]^2 a matrix product: two-term sums per entry, no reassociation possible
            worst[f] = max(worst[f], _ulps(_oracle_prim(f, q, None, 9), _ref_call(ref, f, q, None, 9)))
    assert worst == {"expMapRot": 0.0, "expMapJac": 0.0}, worst
    # logMap: generic rotations, tiny angles (Taylor branch below 1e-6), and angles within 1e-6 of pi (the reference's sqrt branch)
    Rs = [_rot(rng) for _ in range(300)]
    for ang in (0.0, 1e-9, 9e-7, 1.1e-6, 1e-3, np.pi - 1e-3, np.pi - 2e-6, np.pi - 5e-7, np.pi - 1e-9, np.pi):
        for _ in range(20):
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            Rs.append(_ref_call(ref, "expAngular", d * ang, None, 12)[:9].reshape(3, 3))
    near_pi = 0
    for R in Rs:
        got, want = _oracle_prim("logMap", R.reshape(9), None, 3), _ref_call(ref, "logMap", R.reshape(9), None, 3)
        assert np.array_equal(got, want), (R, got, want)
        near_pi += int(np.linalg.norm(want) > np.pi - 1e-6)
    assert near_pi >= 20                         # the branch next to pi was really taken


def test_adjoint_maps_and_transform_inertia_match_the_references_to_a_few_ulps(ref):
    rng = np.random.default_rng(2)
    worst = {}
    for trial in range(500):
        T = _t12(rng, 10.0 if trial % 5 == 0 else 1.0)
        V, W = rng.normal(0, 1, 6), rng.normal(0, 1, 6)
        for f, a, b in (("AdT", T, V), ("AdInvT", T, V), ("dAdT", T, V), ("dAdInvT", T, V), ("ad", V, W), ("dad", V, W), ("AdInvRLinear", T, V[:3])):
            got, want = _oracle_prim(f, a, b, 6), _ref_call(ref, f, a, b, 6)
            worst[f] = max(worst.get(f, 0.0), _ulps(got, want))
        # a physical spatial inertia (Inertia::computeSpatialTensor, Inertia.cpp:1368-1383) moved by T
        G = _spatial_tensor(rng.uniform(0.1, 10), rng.normal(0, 0.2, 3), _spd3(rng))
        got, want = _oracle_prim("transformInertia", T, G.reshape(36), 36), _ref_call(ref, "transformInertia", T, G.reshape(36), 36)
        assert np.array_equal(want.reshape(6, 6), want.reshape(6, 6).T)
        worst["transformInertia"] = max(worst.get("transformInertia", 0.0), float(np.abs(got - want).max() / np.abs(want).max()))
    print("worst disagreement with the reference's functions (ulps; transformInertia: relative):", worst)
    assert worst["ad"] == 0.0 and worst["dad"] == 0.0 and worst["AdInvRLinear"] <= 2.0         # cross products: the same two-term expressions
    for f in ("AdT", "AdInvT", "dAdT", "dAdInvT"):
        assert worst[f] <= 8.0, (f, worst[f])
    # the reference's unrolled closed form (186 multiplications) against the oracle's explicit congruence Ad^T G Ad
    assert worst["transformInertia"] <= 1e-14


def _spd3(rng):
    A = rng.normal(size=(3, 3))
    return A @ A.T * 0.05 + 0.02 * np.eye(3)


def _skew(c):
    return np.array([[0, -c[2], c[1]], [c[2], 0, -c[0]], [-c[1], c[0], 0]])


def _spatial_tensor(m, c, Ic):
    Cx = _skew(c)
    G = np.zeros((6, 6))
    G[:3, :3] = Ic + m * Cx @ Cx.T; G[:3, 3:] = m * Cx; G[3:, :3] = m * Cx.T; G[3:, 3:] = m * np.eye(3)
    return G


def _random_tree(rng, nb):
    """A random tree of one-DOF joints (revolute / prismatic / screw, random joint frames on both sides, random inertias) as a model
    description for the oracle, and the same tree in the terms of the reference's SimpleFeatherstone: the joint's screw axis in the
    CHILD BODY frame (S = Ad(T_cj) s: expMap(S q) = T_cj expMap(s q) T_cj^-1, so T_pj Q(q) T_cj^-1 = (T_pj T_cj^-1) expMap(S q)),
    transformFromParent = T_pj T_cj^-1, transformFromChildren = identity, the body's spatial tensor."""
    bodies, parent, axis6, fromParent, inertia = [], [], [], [], []
    for i in range(nb):
        par = -1 if i == 0 else int(rng.integers(0, i))
        jt = rng.choice(["revolute", "prismatic", "screw"], p=[0.6, 0.25, 0.15])
        a = rng.normal(size=3); a /= np.linalg.norm(a)
        T_pj = na.make_transform(rng.normal(0, 0.3, 3), R=_rot(rng)); T_cj = na.make_transform(rng.normal(0, 0.3, 3), R=_rot(rng))
        m = float(rng.uniform(0.2, 5)); c = rng.normal(0, 0.1, 3); Ic = _spd3(rng)
        pitch = float(rng.uniform(0.05, 0.5))
        kw = {"pitch": pitch} if jt == "screw" else {}
        bodies.append(na.BodySpec(f"b{i}", par, jt, f"j{i}", axis=tuple(a), T_pj=T_pj, T_cj=T_cj, mass=m, com=tuple(c),
                                  inertia=(Ic[0, 0], Ic[1, 1], Ic[2, 2], Ic[0, 1], Ic[0, 2], Ic[1, 2]), **kw))
        s = {"revolute": np.concatenate([a, np.zeros(3)]), "prismatic": np.concatenate([np.zeros(3), a]),
             "screw": np.concatenate([a, a * pitch / (2 * np.pi)])}[jt]
        R, p = T_cj[:3, :3], T_cj[:3, 3]
        S = np.concatenate([R @ s[:3], np.cross(p, R @ s[:3]) + R @ s[3:]])           # Ad(T_cj) s
        F = T_pj @ np.linalg.inv(T_cj)
        parent.append(par); axis6.append(S); fromParent.append(np.concatenate([F[:3, :3].reshape(9), F[:3, 3]]))
        inertia.append(_spatial_tensor(m, c, Ic).reshape(36))
    md = na.ModelDescription("tree", bodies, [], gravity=(0.0, 0.0, 0.0), dt=1e-3, max_contacts=0)
    return md, np.array(parent, np.int32), np.array(axis6), np.array(fromParent), np.array(inertia)


@pytest.mark.parametrize("nb", [1, 2, 5, 12, 30])
def test_oracle_aba_equals_the_references_flat_array_featherstone_on_random_trees(ref, nb):
    """Skeleton::computeForwardDynamics as the oracle restates it (oracle/dynamics.hpp; gravity off, no damping / springs: the
    reference's SimpleFeatherstone has neither) against SimpleFeatherstone::forwardDynamics compiled from the reference."""
    ref.ref_simple_featherstone.argtypes = [C.c_int, C.POINTER(C.c_int32), PD, PD, PD, PD, PD, PD, PD, PD]
    ref.ref_simple_featherstone.restype = None
    ident = np.tile(np.concatenate([np.eye(3).reshape(9), np.zeros(3)]), (nb, 1))
    worst = 0.0
    for seed in range(40):
        rng = np.random.default_rng(1000 * nb + seed)
        md, parent, axis6, fromParent, inertia = _random_tree(rng, nb)
        ow = OracleWorld(md)
        assert ow.n == nb
        q, v, tau = rng.uniform(-np.pi, np.pi, nb), rng.normal(0, 2.0, nb), rng.normal(0, 5.0, nb)
        acc = np.zeros(nb)
        ref.ref_simple_featherstone(nb, parent.ctypes.data_as(C.POINTER(C.c_int32)), _p(np.ascontiguousarray(axis6)), _p(np.ascontiguousarray(fromParent)),
                                    _p(np.ascontiguousarray(ident)), _p(np.ascontiguousarray(inertia)), _p(q), _p(v), _p(tau), _p(acc))
        mine = ow.forward_dynamics(q, v, tau)
        assert np.all(np.isfinite(acc)) and np.abs(acc).max() > 0
        worst = max(worst, float(np.abs(mine - acc).max() / np.abs(acc).max()))
    print(f"{nb} bodies: oracle ABA vs the reference's SimpleFeatherstone, worst relative difference over 40 random trees: {worst:.2e}")
    assert worst <= 1e-14


def test_free_joint_position_integration_is_bit_identical_to_the_references(ref):
    """FreeJoint::integratePositionsExplicit (FreeJoint.cpp:922-929, the DART_USE_IDENTITY_JACOBIAN branch the reference ships):
    q' = log(T(q) T(v dt)) - row a6 - against the oracle's integratePositions on a single free body, including rotations that land next
    to pi (the sqrt branch of logMap) and steps below the Taylor thresholds."""
    ref.ref_free_joint_integrate.argtypes = [PD, PD, C.c_double, PD]
    ref.ref_free_joint_integrate.restype = None
    rng = np.random.default_rng(3)
    worst = 0.0
    for dt in (1e-3, 5e-3, 1e-5):
        md = na.ModelDescription("free", [na.BodySpec("b", -1, "free", "j", mass=1.0, inertia=(0.1, 0.1, 0.1, 0, 0, 0))], [], gravity=(0, -9.81, 0), dt=dt, max_contacts=0)
        ow = OracleWorld(md)
        for trial in range(300):
            q = np.concatenate([rng.normal(0, [0.01, 1.0, 2.5][trial % 3], 3), rng.normal(0, 2, 3)])
            v = np.concatenate([rng.normal(0, [1e-4, 1.0, 30.0][(trial // 3) % 3], 3), rng.normal(0, 3, 3)])
            if trial % 25 == 0:                      # land within 1e-7 of a rotation by pi
                d = rng.normal(size=3); d /= np.linalg.norm(d)
                q[:3] = d * (np.pi - 1e-7 * rng.uniform(0, 1)); v[:3] = d * 1e-9 / dt
            want = np.zeros(6)
            ref.ref_free_joint_integrate(_p(q), _p(v), dt, _p(want))
            got = ow.integrate_positions(q, v)
            worst = max(worst, _ulps(got, want))
    print("free-joint integration vs the reference's, worst difference (ulps):", worst)
    assert worst <= 4.0          # R(q) R(v dt) is a 3 x 3 product (three-term sums); everything else is element-wise


def test_contact_tangent_basis_is_bit_identical_to_the_references(ref):
    """ContactConstraint::getTangentBasisMatrixODE (ContactConstraint.cpp:734-795, row a8) including its fallback branches (normal along
    the first frictional direction z, then x)."""
    rng = np.random.default_rng(4)
    normals = [rng.normal(size=3) for _ in range(300)]
    normals += [np.array([0.0, 0.0, 1.0]), np.array([0.0, 0.0, -1.0]), np.array([0, 1e-7, 1.0]), np.array([1e-7, 0, -1.0]), np.array([0.0, 1.0, 0.0]),
                np.array([1.0, 0.0, 0.0]), np.array([5e-7, 5e-7, 1.0]), np.array([2e-6, 0, 1.0])]
    for n in normals:
        n = n / np.linalg.norm(n)
        got = _oracle_prim("tangentBasis", n, None, 6)
        want = np.zeros(6)
        ref.ref_tangent_basis.argtypes = [PD, PD]; ref.ref_tangent_basis.restype = None
        ref.ref_tangent_basis(_p(np.ascontiguousarray(n)), _p(want))
        assert np.array_equal(got, want), (n, got, want)


def test_contact_geometry_gradients_match_the_references(ref):
    """The scalar-parameter derivatives behind row a18: ContactConstraint::getTangentBasisMatrixODEGradient (ContactConstraint.cpp:800-876)
    and math::getContactPointGradient (Geometry.cpp:1129-1236, the edge-edge contact point with both radii 1 as
    DifferentiableContactConstraint calls it)."""
    ref.ref_tangent_basis_gradient.argtypes = [PD, PD, PD]; ref.ref_tangent_basis_gradient.restype = None
    ref.ref_getContactPointGradient.argtypes = [PD, C.c_double, C.c_double, PD]; ref.ref_getContactPointGradient.restype = None
    rng = np.random.default_rng(5)
    worst_t = worst_p = 0.0
    normals = [rng.normal(size=3) for _ in range(300)] + [np.array([0.0, 0.0, 1.0]), np.array([0, 1e-7, -1.0]), np.array([0.0, 1.0, 0.0]), np.array([1.0, 0, 0])]
    for n in normals:
        n = np.ascontiguousarray(n / np.linalg.norm(n)); g = rng.normal(0, 1, 3)
        want = np.zeros(6)
        ref.ref_tangent_basis_gradient(_p(n), _p(g), _p(want))
        worst_t = max(worst_t, _ulps(_oracle_prim("tangentBasisGradient", n, g, 6), want))
    for trial in range(500):
        x = rng.normal(0, 1, 24)
        for k in (6, 18):                                     # unit edge directions
            x[k:k + 3] /= np.linalg.norm(x[k:k + 3])
        if trial % 50 == 0:
            x[18:21] = x[6:9]                                  # parallel edges: the d <= 0 branch
        want = np.zeros(3)
        ref.ref_getContactPointGradient(_p(x), 1.0, 1.0, _p(want))
        worst_p = max(worst_p, _ulps(_oracle_prim("contactPointGradient", x, None, 3), want))
    # capsule contacts (PIPE_A / PIPE_B, DCC.cpp:510-547, 862-938) call it with the capsules' normalised radii and with (0, 1) / (1, 0);
    # SPHERE_TO_PIPE / PIPE_TO_SPHERE go through math::closestPointOnLineGradient (Geometry.cpp:4427-4445)
    ref.ref_closestPointOnLineGradient.argtypes = [PD, PD]; ref.ref_closestPointOnLineGradient.restype = None
    worst_r = worst_l = 0.0
    for trial in range(600):
        x = rng.normal(0, 1, 24)
        for k in (6, 18):
            x[k:k + 3] /= np.linalg.norm(x[k:k + 3])
        if trial % 50 == 0:
            x[18:21] = x[6:9]
        rA = rng.uniform(0.05, 0.95)
        radii = np.array([(rA, 1 - rA), (0.0, 1.0), (1.0, 0.0)][trial % 3])
        want = np.zeros(3)
        ref.ref_getContactPointGradient(_p(x), radii[0], radii[1], _p(want))
        worst_r = max(worst_r, _ulps(_oracle_prim("contactPointGradientRadii", x, radii, 3), want))
        y = rng.normal(0, 1, 18); y[6:9] /= np.linalg.norm(y[6:9])
        ref.ref_closestPointOnLineGradient(_p(y), _p(want))
        worst_l = max(worst_l, _ulps(_oracle_prim("closestPointOnLineGradient", y, None, 3), want))
    print("contact-point gradient with radii / closest-point-on-line gradient vs the reference's (ulps):", worst_r, worst_l)
    assert worst_r == 0.0 and worst_l == 0.0
    print("tangent-basis gradient / contact-point gradient vs the reference's, worst difference (ulps):", worst_t, worst_p)
    assert worst_t == 0.0 and worst_p == 0.0       # (the pin found the oracle multiplying by 1 / |t| where the reference divides: 16 ulps, fixed)
