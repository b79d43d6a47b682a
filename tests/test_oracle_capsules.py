"""Capsule colliders in the CPU oracle: the closed-form narrow phases collideCapsuleCapsule / collideSphereCapsule / collideCapsuleSphere
(DARTCollide.cpp:4183-4420) against (a) the scenarios and expected values of the reference's own unit tests
(unittests/unit/test_DARTCollide.cpp:2166-2560), (b) the reference's functions themselves compiled from its file
(oracle/_ref/libdboxbox_ref.so), and the gradient model of the capsule contact types (SPHERE_PIPE / PIPE_SPHERE / PIPE_PIPE,
DCC.cpp:484-547, 819-938) pinned the way the reference pins its gradients: VJP == J^T g with J by central differences of the step."""
import ctypes as C
import os

import numpy as np
import pytest

import nimblephysics_amd as na
import oracle
from oracle import OracleWorld
from test_oracle_spheres import _fd_check
from util import capsule_world

L = oracle._lib()
L.nbo_capsule_pair.restype = C.c_int
PIPE_SPHERE, SPHERE_PIPE, PIPE_PIPE, SPHERE_SPHERE = 13, 14, 15, 6


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _T(R=None, p=(0, 0, 0)):
    return np.ascontiguousarray(np.concatenate([(np.eye(3) if R is None else R).reshape(9), np.asarray(p, dtype=np.float64)]))


def _rot_y(a):   # math::eulerXYZToMatrix((0, a, 0))
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _pair(which, T1, size1, T2, size2, clip=0.03):
    out = np.zeros(48)
    n = L.nbo_capsule_pair(which, _p(T1), _p(np.ascontiguousarray(size1, dtype=np.float64)), _p(T2), _p(np.ascontiguousarray(size2, dtype=np.float64)),
                           C.c_double(clip), _p(out))
    return n, out


# (name, which, T1, size1 (radius, height), T2, size2, expected type, expected point, expected normal) - and the same pair in the
# other order with the mirrored type and the opposite normal, exactly as the reference's tests run them
H, R1, R2 = 1.0, 0.4, 0.3
_S = np.sqrt(2) * H / 4
REFERENCE_TESTS = [
    ("CAPSULE_CAPSULE_T_SHAPED", 0, _T(), (R1, H), _T(_rot_y(np.pi / 2), (R1 + R2 + H / 2 - 0.01, 0, 0)), (R2, H),
     PIPE_SPHERE, SPHERE_PIPE, np.array([1.0, 0, 0]) * (R1 - 0.01 * R1 / (R1 + R2)), np.array([-1.0, 0, 0])),
    ("CAPSULE_CAPSULE_X_SHAPED", 0, _T(), (R1, H), _T(_rot_y(np.pi / 2), (0, R1 + R2 - 0.01, 0)), (R2, H),
     PIPE_PIPE, PIPE_PIPE, np.array([0, 1.0, 0]) * (R1 - 0.01 * R1 / (R1 + R2)), np.array([0, -1.0, 0])),
    ("CAPSULE_CAPSULE_L_SHAPED", 0, _T(), (R1, H), _T(_rot_y(np.pi / 4), (_S, 0, H / 2 + _S + R1 + R2 - 0.01)), (R2, H),
     SPHERE_SPHERE, SPHERE_SPHERE, np.array([0, 0, 1.0]) * (H / 2 + R1 - 0.01 * R1 / (R1 + R2)), np.array([0, 0, -1.0])),
    ("CAPSULE_SPHERE_END", 2, _T(), (R1, H), _T(None, (0, 0, H / 2 + R1 + R2 - 0.01)), (R2, 0),
     SPHERE_SPHERE, SPHERE_SPHERE, np.array([0, 0, 1.0]) * (H / 2 + R1 - 0.01 * R1 / (R1 + R2)), np.array([0, 0, -1.0])),
    ("CAPSULE_SPHERE_SIDE", 2, _T(), (R1, H), _T(None, (R1 + R2 - 0.01, 0, 0)), (R2, 0),
     PIPE_SPHERE, SPHERE_PIPE, np.array([1.0, 0, 0]) * (R1 - 0.01 * R1 / (R1 + R2)), np.array([-1.0, 0, 0])),
]


@pytest.mark.parametrize("case", REFERENCE_TESTS, ids=[c[0] for c in REFERENCE_TESTS])
def test_the_scenarios_of_the_references_own_capsule_tests(case):
    name, which, T1, s1, T2, s2, type_fwd, type_bwd, point, normal = case
    n, c = _pair(which, T1, s1, T2, s2)
    assert n == 1 and int(c[7]) == type_fwd
    assert np.abs(c[0:3] - point).max() < 1e-10 and np.abs(c[3:6] - normal).max() < 1e-10 and abs(c[6] - 0.01) < 1e-10
    # "Check the results in the backwards direction"
    n, c = _pair({0: 0, 2: 1}[which], T2, s2, T1, s1)
    assert n == 1 and int(c[7]) == type_bwd
    assert np.abs(c[0:3] - point).max() < 1e-10 and np.abs(c[3:6] + normal).max() < 1e-10 and abs(c[6] - 0.01) < 1e-10


def _ref():
    path = os.path.join(os.path.dirname(oracle.__file__), "_ref", "libdboxbox_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libdboxbox_ref.so not built (needs /root/reference at build time)")
    ref = C.CDLL(path)
    ref.ref_collide_capsule.restype = C.c_int
    return ref


def test_capsule_narrow_phases_equal_the_references_compiled_functions():
    """The reference's collideCapsuleCapsule / collideSphereCapsule / collideCapsuleSphere with dSegmentsClosestApproach /
    dDistPointToSegment, compiled from dart/collision/dart/DARTCollide.cpp where it lies (oracle/ref_build.py), next to the oracle's
    restatement on random pairs: crossing axes, parallel axes, end against side, end against end, spheres against sides and caps,
    touching, too deep, separated.  Same count and contact type; every number of the contact BIT for bit."""
    ref = _ref()
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(123)
    seen = {}
    for trial in range(12000):
        which = trial % 3
        r0, r1 = rng.uniform(0.05, 0.4, 2)
        h0, h1 = rng.uniform(0.1, 1.2, 2)
        Ra = Rotation.random(random_state=rng.integers(1 << 31)).as_matrix()
        pa = rng.normal(0, 1, 3)
        # a point on / next to the axis of the first capsule, the second shape placed around it at about the touching distance
        t = rng.choice([rng.uniform(-0.5, 0.5), -0.5, 0.5, rng.uniform(-0.7, 0.7)])
        on_axis = pa + Ra[:, 2] * t * h0
        d = rng.normal(0, 1, 3)
        if trial % 5:
            d -= Ra[:, 2] * (d @ Ra[:, 2])                      # sideways: pipe contacts
        d /= np.linalg.norm(d)
        gap = (r0 + r1) * rng.choice([rng.uniform(0.93, 1.0), rng.uniform(0.99, 1.02), rng.uniform(0.5, 0.9)])
        if which == 0:
            Rb = Rotation.random(random_state=rng.integers(1 << 31)).as_matrix()
            if trial % 11 == 0:
                Rb = Ra.copy()                                  # parallel axes: the D < SMALL_NUM branch
            if trial % 7 == 0:
                Rb = Ra @ _rot_y(np.pi / 2)                     # T / L shapes
            s = rng.choice([rng.uniform(-0.5, 0.5), -0.5, 0.5])
            pb = on_axis + d * gap - Rb[:, 2] * s * h1
            args = (_T(Ra, pa), (r0, h0, 0), _T(Rb, pb), (r1, h1, 0))
        else:
            cs = on_axis + d * gap
            caps, sph = (_T(Ra, pa), (r0, h0, 0)), (_T(None, cs), (r1, r1, r1))
            args = sph + caps if which == 1 else caps + sph
        no, o = _pair(which, *args)
        r = np.zeros(48)
        nr = ref.ref_collide_capsule(which, _p(np.ascontiguousarray(args[1], dtype=np.float64)), _p(args[0]),
                                     _p(np.ascontiguousarray(args[3], dtype=np.float64)), _p(args[2]), C.c_double(0.03), _p(r), 1)
        assert no == nr, (trial, which, no, nr)
        if nr:
            assert np.array_equal(o, r), (trial, which, o[7], r[7], np.abs(o - r).max())
        key = (which, int(r[7]) if nr else 0)
        seen[key] = seen.get(key, 0) + 1
    print("capsule narrow phases vs the reference's compiled functions, (which, type): count =", dict(sorted(seen.items())))
    for key in [(0, PIPE_PIPE), (0, PIPE_SPHERE), (0, SPHERE_PIPE), (0, SPHERE_SPHERE), (1, SPHERE_PIPE), (1, SPHERE_SPHERE),
                (2, PIPE_SPHERE), (2, SPHERE_SPHERE), (0, 0), (1, 0), (2, 0)]:
        assert seen.get(key, 0) > 20, (key, seen)


# ---- gradients: VJP == J^T g ----------------------------------------------------------------------------------------------------
def _state(md, poses, seed, vel=0.02):
    """poses: per free body (rotation vector, position)"""
    rng = np.random.default_rng(seed)
    n = md.num_dofs
    q = np.zeros(n)
    for i, (rv, p) in enumerate(poses):
        q[6 * i:6 * i + 3] = rv; q[6 * i + 3:6 * i + 6] = p
    return np.concatenate([q, rng.normal(0, vel, n)]), rng.normal(0, 0.1, n)


def test_crossed_capsules_pipe_pipe_contact_and_gradient():
    """A free capsule lying across a world-fixed capsule: PIPE_PIPE, both orders of the pair."""
    for order in ("fixed_first", "free_first"):
        md = capsule_world(order=order, kinds=("capsule",))
        s0, a0 = _state(md, [((0.0, np.pi / 2 + 0.2, 0.1), (0.03, 0.25 + 0.1 - 0.004, 0.02))], 1)
        w = _fd_check(md, s0, a0, 2, tol=5e-6)
        c = w.last_contacts()
        assert c.shape[0] == 1 and int(c[0, 7]) == PIPE_PIPE and 0 < c[0, 6] < 0.01


def test_capsule_end_on_capsule_side_gradient():
    """The end cap of a free capsule standing on the side of the fixed capsule: SPHERE_PIPE / PIPE_SPHERE by pair order."""
    for order, want in (("fixed_first", PIPE_SPHERE), ("free_first", SPHERE_PIPE)):
        md = capsule_world(order=order, kinds=("capsule",))
        # axis of the free capsule (its z) turned towards -y (a rotation about x by pi/2 + 0.15), its lower end 4 mm inside
        th = np.pi / 2 + 0.15
        axis = np.array([0.0, -np.sin(th), np.cos(th)])
        radial = np.array([0.1, 1.0, 0.0]) / np.linalg.norm([0.1, 1.0])
        end = radial * (0.25 + 0.1 - 0.004) + np.array([0, 0, 0.05])
        s0, a0 = _state(md, [((th, 0.0, 0.0), end - 0.2 * axis)], 3)
        w = _fd_check(md, s0, a0, 4, tol=5e-6)
        c = w.last_contacts()
        assert c.shape[0] == 1 and int(c[0, 7]) == want, c


def test_sphere_on_capsule_side_and_capsule_on_sphere_gradient():
    for order, want in (("fixed_first", PIPE_SPHERE), ("free_first", SPHERE_PIPE)):
        md = capsule_world(order=order, kinds=("sphere",))
        radial = np.array([0.2, 1.0, 0.0]) / np.linalg.norm([0.2, 1.0])
        s0, a0 = _state(md, [((0.3, -0.2, 0.1), radial * (0.25 + 0.1 - 0.003) + np.array([0, 0, 0.04]))], 5)
        w = _fd_check(md, s0, a0, 6, tol=5e-6)
        c = w.last_contacts()
        assert c.shape[0] == 1 and int(c[0, 7]) == want, c


def test_two_free_capsules_and_a_sphere_on_the_fixed_capsule():
    """Three bodies: a capsule across the fixed one, a second capsule leaning on the first with its end cap, a sphere against the side of
    the first - every capsule contact type in one world, two of the contacts between two MOVING bodies."""
    md = capsule_world(order="fixed_first", kinds=("capsule", "capsule", "sphere"))
    poses = [((0.0, np.pi / 2, 0.0), (0.0, 0.346, 0.0)),                       # across the fixed capsule (its axis along x)
             ((np.pi / 2, 0.0, 0.0), (0.15, 0.346 + 0.1 + 0.1 + 0.2 - 0.003, 0.0)),   # standing on the first one (axis along -y)
             ((0.0, 0.0, 0.0), (-0.15, 0.346 + 0.1 + 0.1 - 0.003, 0.0))]          # a ball on the first one
    s0, a0 = _state(md, poses, 7)
    w = _fd_check(md, s0, a0, 8, tol=1e-5)
    types = sorted(int(t) for t in w.last_contacts()[:, 7])
    assert types == [PIPE_SPHERE, PIPE_SPHERE, PIPE_PIPE], types


def test_a_capsule_that_can_meet_a_box_is_refused():
    md = capsule_world(order="fixed_first", kinds=("capsule",))
    md.boxes.append(na.BoxSpec(-1, na.make_transform((0, -2, 0)), (1, 1, 1), 1.0))
    with pytest.raises(Exception):
        OracleWorld(md)
