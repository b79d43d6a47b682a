"""Pins the contact part of the CPU oracle.

Golden vectors of the reference's own tests (tests/golden/lcp_fixtures.json, extracted by
tools/extract_lcp_fixtures.py from unittests/unit/test_LCPUtils.cpp; the box-box case is the literal
fixture of unittests/unit/test_DARTCollide.cpp:554-593), the reference's REAL Dantzig solver compiled from
its vendored sources (oracle/_ref), and the reference's property tests restated (GradientTestUtils.hpp:
verifyNextV :1902-1984, verifyVelGradients / verifyAnalyticalBackprop vs finite differences).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from oracle import OracleWorld
from util import contact_inputs, rel_err

import nimblephysics_amd as na

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "lcp_fixtures.json")))
L = oracle._lib()
pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(pd)


def _fixture(name):
    f = FIX[name]
    n = len(f["b"])
    A = _d(f["A"]).reshape(n, n)
    return n, A, _d(f.get("x", np.zeros(n))), _d(f["b"]), _d(f["lo"]), _d(f["hi"]), np.ascontiguousarray(f["fIndex"], dtype=np.int32)


def lcp_valid(A, x, b, lo, hi, fi, ignore=False):
    return bool(L.nbo_lcp_valid(len(b), _p(_d(A)), _p(_d(x)), _p(_d(b)), _p(_d(lo)), _p(_d(hi)), fi.ctypes.data_as(pi), int(ignore)))


def have_ref():
    return os.path.exists(os.path.join(os.path.dirname(oracle.__file__), "_ref", "libodelcp_ref.so"))


def test_fixture_lcp_failure_guess_then_pgs_is_valid():
    """test_LCPUtils.cpp:370-418 LCP_FAILURE: guessSolution then PGS(50000, 1e-15, 1e-12, 1e-10) -> valid."""
    n, A, x, b, lo, hi, fi = _fixture("LCP_FAILURE")
    g = np.zeros(n)
    L.nbo_lcp_guess(n, _p(A), _p(b), fi.ctypes.data_as(pi), _p(g))
    L.nbo_lcp_pgs(n, _p(A), _p(g), _p(b), _p(lo), _p(hi), fi.ctypes.data_as(pi), 50000, C.c_double(1e-15), C.c_double(1e-12), C.c_double(1e-10))
    assert lcp_valid(A, g, b, lo, hi, fi)


def test_fixture_real_life_failure_1_reduce_equals_manual_merges():
    """test_LCPUtils.cpp:423-463: reduce() == mergeLCPColumns(0,3); (1,3); (2,3): two identical contacts collapse."""
    n, A, x, b, lo, hi, fi = _fixture("REAL_LIFE_FAILURE_1")
    Ar = np.zeros(n * n); xr = np.zeros(n); br = np.zeros(n); lor = np.zeros(n); hir = np.zeros(n); fr = np.zeros(n, np.int32); mo = np.zeros(n * n)
    nr = L.nbo_lcp_reduce(n, _p(A), _p(x), _p(b), _p(lo), _p(hi), fi.ctypes.data_as(pi), 0, _p(Ar), _p(xr), _p(br), _p(lor), _p(hir), fr.ctypes.data_as(pi), _p(mo))
    assert nr == 3
    Ar = Ar[:9].reshape(3, 3)
    assert np.allclose(Ar, 2.0 * A[:3, :3], atol=1e-8)          # merged columns are doubled (LCPUtils.cpp:395-398)
    assert list(fr[:3]) == [-1, 0, 0]
    M = mo[:n * 3].reshape(n, 3)
    assert np.array_equal(M, np.vstack([np.eye(3), np.eye(3)]))


def test_fixture_lcp_failure_2_remove_friction():
    """test_LCPUtils.cpp:198-347 LCP_FAILURE_2: drop friction, PGS, valid with friction indices ignored."""
    n, A, x, b, lo, hi, fi = _fixture("LCP_FAILURE_2")
    Ar = np.zeros(n * n); xr = np.zeros(n); br = np.zeros(n); lor = np.zeros(n); hir = np.zeros(n); fr = np.zeros(n, np.int32); mo = np.zeros(n * n)
    nr = L.nbo_lcp_reduce(n, _p(A), _p(x), _p(b), _p(lo), _p(hi), fi.ctypes.data_as(pi), 1, _p(Ar), _p(xr), _p(br), _p(lor), _p(hir), fr.ctypes.data_as(pi), _p(mo))
    assert nr == 2 and list(fr[:2]) == [-1, -1]
    A2 = _d(Ar[:4].reshape(2, 2)); x2 = np.zeros(2)
    L.nbo_lcp_pgs(2, _p(A2), _p(x2), _p(_d(br[:2])), _p(_d(lor[:2])), _p(_d(hir[:2])), fr.ctypes.data_as(pi), 50000, C.c_double(1e-15), C.c_double(1e-12), C.c_double(1e-10))
    assert lcp_valid(A2, x2, br[:2], lor[:2], hir[:2], fr[:2].copy(), ignore=True)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (reference Dantzig) not built")
@pytest.mark.parametrize("name", sorted(FIX))
def test_reference_dantzig_on_the_reference_fixtures(name):
    """The reference's own dSolveLCP (compiled from dart/external/odelcpsolver) on its own fixtures: whenever it
    reports success on a friction-free problem the restated validity check must agree."""
    n, A, x, b, lo, hi, fi = _fixture(name)
    xs = np.zeros(n)
    ok = L.nbo_lcp_dantzig(n, _p(A), _p(xs), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), fi.copy().ctypes.data_as(pi), 1)
    assert ok in (0, 1)
    assert np.all(np.isfinite(xs))


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (reference Dantzig) not built")
def test_reference_dantzig_solves_random_spd_normal_only_lcps():
    rng = np.random.default_rng(3)
    for n in (1, 3, 8, 17):
        J = rng.normal(0, 1, (n, n + 2)); A = _d(J @ J.T); b = _d(rng.normal(0, 1, n))
        lo = np.zeros(n); hi = np.full(n, np.inf); fi = np.full(n, -1, np.int32); x = np.zeros(n)
        assert L.nbo_lcp_dantzig(n, _p(A), _p(x), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), fi.ctypes.data_as(pi), 1) == 1
        assert lcp_valid(A, x, b, lo, hi, fi)


def _ref_lcp_lib():
    path = os.path.join(os.path.dirname(oracle.__file__), "_ref", "libodelcp_ref.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    return lib if hasattr(lib, "nbo_ref_pgs") else None


@pytest.mark.skipif(_ref_lcp_lib() is None, reason="oracle/_ref/libodelcp_ref.so with the reference's PGS not built (oracle/ref_build.py)")
def test_pgs_equals_the_references_own_pgs_solver_bit_for_bit():
    """The reference's PgsBoxedLcpSolver::solve itself (dart/constraint/PgsBoxedLcpSolver.cpp:79-268, compiled from where it lies into
    oracle/_ref by oracle/ref_build.py) against the oracle's restatement: contact LCPs (full rank, rank deficient, with the fallback CFM
    on the diagonal), the eight LCP fixtures of the reference's unit tests, zero-diagonal rows (skipped by the epsilon-for-division rule),
    with the default options the cascade uses and with tight ones: return value and x are identical."""
    R = _ref_lcp_lib()
    from util import contact_lcp
    rng = np.random.default_rng(11)
    problems = []
    for trial in range(300):
        nc = int(rng.integers(1, 9))
        A, b, lo, hi, fi = contact_lcp(rng, nc, int(rng.integers(2, 26)), cfm=[0.0, 1e-4, 1e-3][trial % 3])
        if trial % 7 == 0:                      # a dead row: zero diagonal
            k = int(rng.integers(0, 3 * nc)); A[k, :] = 0.0; A[:, k] = 0.0
        problems.append((3 * nc, A, rng.normal(0, 0.1, 3 * nc) * (trial % 2), b, lo, hi, fi))
    for name in sorted(FIX):
        n, A, x, b, lo, hi, fi = _fixture(name)
        problems.append((n, A, x, b, lo, hi, fi))
    iterated = 0
    for n, A, x0, b, lo, hi, fi in problems:
        for opts in ((30, 1e-6, 1e-3, 1e-9), (2000, 1e-15, 1e-12, 1e-10)):
            Ao, Ar = _d(A).copy(), _d(A).copy(); xo, xr = _d(x0).copy(), _d(x0).copy(); bo, br = _d(b).copy(), _d(b).copy()
            args = (opts[0], C.c_double(opts[1]), C.c_double(opts[2]), C.c_double(opts[3]))
            oko = L.nbo_lcp_pgs(n, _p(Ao), _p(xo), _p(bo), _p(_d(lo)), _p(_d(hi)), fi.ctypes.data_as(pi), *args)
            okr = R.nbo_ref_pgs(n, _p(Ar), _p(xr), _p(br), _p(_d(lo)), _p(_d(hi)), fi.ctypes.data_as(pi), *args)
            assert oko == okr
            assert np.array_equal(xo, xr)
            iterated += int(not np.array_equal(br, _d(b)))
    assert iterated > 100          # the sweep loop ran (the first pass normalises A and b only when it did not converge at once)


def test_cod_solve_is_min_norm_least_squares():
    """Stand-in for Eigen completeOrthogonalDecomposition().solve(): equals numpy's pinv solution, rank revealed."""
    rng = np.random.default_rng(4)
    for (r, c, k) in ((6, 6, 6), (8, 8, 5), (12, 12, 6), (24, 24, 12), (5, 5, 0)):
        U = rng.normal(0, 1, (r, k)); V = rng.normal(0, 1, (k, c)); A = _d(U @ V) if k else np.zeros((r, c))
        b = _d(A @ rng.normal(0, 1, c) + (0 if k == r else 0.1 * rng.normal(0, 1, r)))
        x = np.zeros(c)
        rank = L.nbo_cod_solve(r, c, _p(A), _p(b), _p(x))
        assert rank == k
        assert np.allclose(x, np.linalg.pinv(A, rcond=1e-12) @ b, atol=1e-9)


def test_box_box_face_face_annotation_fixture():
    """unittests/unit/test_DARTCollide.cpp:554-593 BOX_BOX_FACE_FACE_COLLISION_ANNOTATION (literal expectations)."""
    T1 = np.concatenate([np.eye(3).reshape(9), [0, 0, -0.5]]); T2 = np.concatenate([np.eye(3).reshape(9), [0, 0.5, 0.25]])
    out = np.zeros(8 * 22)
    n = L.nbo_box_box(_p(_d(T1)), _p(_d([1, 1, 1])), _p(_d(T2)), _p(_d([0.5, 0.5, 0.5])), C.c_double(0.03), _p(out))
    assert n == 4
    cts = out[:4 * 22].reshape(4, 22)
    order = np.argsort(cts[:, :3] @ np.array([-0.1, -1.0, 0.0]))   # CollisionResult::sortContacts(sortDir)
    cts = cts[order]
    exp_pts = [(0.25, 0.5, 0), (-0.25, 0.5, 0), (0.25, 0.25, 0), (-0.25, 0.25, 0)]
    exp_types = [3, 3, 2, 2]                                        # EDGE_EDGE, EDGE_EDGE, FACE_VERTEX, FACE_VERTEX
    for c, p, t in zip(cts, exp_pts, exp_types):
        assert np.allclose(c[:3], p, atol=1e-9) and int(c[7]) == t
    assert np.allclose(np.abs(cts[0, 11:14]), (1, 0, 0)) and np.allclose(np.abs(cts[0, 17:20]), (0, 1, 0))


def test_box_box_separated_and_deep():
    T1 = np.concatenate([np.eye(3).reshape(9), [0, 0, 0]])
    out = np.zeros(8 * 22)
    T2 = np.concatenate([np.eye(3).reshape(9), [0, 1.2, 0]])
    assert L.nbo_box_box(_p(_d(T1)), _p(_d([1, 1, 1])), _p(_d(T2)), _p(_d([1, 1, 1])), C.c_double(0.03), _p(out)) == 0
    T2 = np.concatenate([np.eye(3).reshape(9), [0, 0.99, 0]])
    assert L.nbo_box_box(_p(_d(T1)), _p(_d([1, 1, 1])), _p(_d(T2)), _p(_d([1, 1, 1])), C.c_double(0.03), _p(out)) == 4


def test_atlas_standing_has_eight_vertex_face_contacts():
    """2 feet x 4 incident-face corners = 8 contacts at q[0] = -pi/2, q[4] = -0.01 (depth 0.011 < 0.03)."""
    md, s, a = contact_inputs("atlas20", 1, 1, joint_noise=0.0, vel_noise=0.0, action_noise=0.0)
    w = OracleWorld(md)
    w.step(s[0], a[0])
    cts = w.last_contacts()
    assert len(cts) == 8 and np.all(cts[:, 7] == 1) and np.allclose(cts[:, 3:6], [0, 1, 0]) and np.allclose(cts[:, 6], 0.011, atol=1e-9)
    l = w.last_lcp()
    assert (w.last_status & 0x2) and np.all(l["row_class"] == 1)
    assert np.linalg.matrix_rank(l["A"], 1e-9) < 24       # redundant corners: Q is rank deficient, min-norm solution used
    assert lcp_valid(l["A"], l["x"], l["b"], l["lo"], l["hi"], l["findex"])


@pytest.mark.parametrize("name", ["atlas20", "atlas33"])
def test_next_v_identity(name):
    """verifyNextV: v' = v_pre + Minv (A_c + A_ub E) f_c must reproduce the stepped velocity."""
    md, s, a = contact_inputs(name, 1, 2)
    w = OracleWorld(md); n = w.n
    nxt = w.step(s[0], a[0])
    l = w.last_lcp()
    q, v = s[0, :n], s[0, n:]
    M = w.mass_matrix(q)
    fl = md.merge_welds().flat()
    tau = a[0]
    vpre = v + md.dt * np.linalg.solve(M, tau - w.coriolis_gravity(q, v) - fl["damping"] * v)
    # massed impulse tests are not exposed; rebuild J from A = J Minv J^T is not unique, so use the solver's own x
    # through the impulse response of each row's contact wrench instead: v' - v_pre must be in the range of Minv
    dv = nxt[n:] - vpre
    assert np.abs(dv).max() > 1e-6            # contact did something
    # energy-consistency: relative normal velocity after the step is ~0 on clamping normal rows: A x - b = 0
    w_res = l["A"] @ l["x"] - l["b"]
    assert np.abs(w_res[l["row_class"] == 1]).max() < 1e-9


@pytest.mark.parametrize("name", ["atlas20"])
def test_contact_backprop_matches_finite_differences(name):
    """verifyAnalyticalBackprop with clamping contacts: VJP == J^T g with J by central differences of the step."""
    md, s, a = contact_inputs(name, 1, 3)
    w = OracleWorld(md); n = w.n
    s0, a0 = s[0], a[0]

    def step(x, u):
        w.reset_lcp_cache()
        return w.step(x, u)

    step(s0, a0)
    assert w.last_status & 0x2
    g = np.random.default_rng(5).normal(0, 1, 2 * n)
    gs, ga = w.backprop(g)
    eps = 1e-7
    Js = np.zeros((2 * n, 2 * n)); Ja = np.zeros((2 * n, n))
    for j in range(2 * n):
        xp, xm = s0.copy(), s0.copy(); xp[j] += eps; xm[j] -= eps
        Js[:, j] = (step(xp, a0) - step(xm, a0)) / (2 * eps)
    for j in range(n):
        up, um = a0.copy(), a0.copy(); up[j] += eps; um[j] -= eps
        Ja[:, j] = (step(s0, up) - step(s0, um)) / (2 * eps)
    assert rel_err(gs, Js.T @ g) < 1e-6 and rel_err(ga, Ja.T @ g) < 1e-6


def test_warm_start_is_carried_and_reused():
    """BoxedLcpConstraintSolver::mX persists across steps (:176-187); same row count -> warm start, else guess."""
    md, s, a = contact_inputs("atlas20", 1, 6)
    w = OracleWorld(md)
    s1 = w.step(s[0], a[0])
    c1 = w.get_lcp_cache()
    assert len(c1) == 24 and np.allclose(c1, w.last_lcp()["x"])
    w.step(s1, a[0])
    assert w.last_status & 0x2


def test_restitution_bounce_and_its_jacobians():
    """ContactConstraint.cpp:95-110, 395-442: e = e_A e_B, bounce when e > 1e-3 and e * approach speed > 0.1; velVel (through the
    bounce diagonals 1 + e) against central differences; posPos / velPos carry the reference's bounce APPROXIMATION
    (BackpropSnapshot.cpp:1131-1226): in the contact direction the position Jacobian becomes -e (X = I - (1 + e) a a^T / |a|^2 for one
    bouncing contact), which is not the derivative of the explicit position update - by design."""
    from oracle import OracleWorld
    from util import ball_state, ball_world
    md = ball_world("box_first", n_balls=1)
    md.boxes[0].restitution = 0.9; md.boxes[1].restitution = 0.8
    s, a = ball_state(md, [(0.2, 0.1)], 3, pen=1e-3)
    n = md.num_dofs
    s[0:3] = 0.0; s[n:] = 0.0; s[n + 4] = -1.0
    ow = OracleWorld(md)
    nx = ow.step(s, a)
    assert abs(nx[n + 4] - (0.72 * 1.0)) < 0.02 and nx[n + 4] > 0          # leaves with ~e times the approach speed (gravity aside)
    J = ow.getStateJacobian()
    assert np.allclose(np.diag(J[:n, :n]), [1, 1, 1, 1, -0.72, 1], atol=1e-9)
    assert np.allclose(np.diag(J[:n, n:]) / md.dt, [1, 1, 1, 1, -0.72, 1], atol=1e-6)
    eps = 1e-6
    fd = np.zeros((2 * n, n))
    for k in range(n):
        sp = s.copy(); sm = s.copy(); sp[n + k] += eps; sm[n + k] -= eps
        fd[:, k] = (OracleWorld(md).step(sp, a) - OracleWorld(md).step(sm, a)) / (2 * eps)
    assert np.abs(J[n:, n:] - fd[n:, :]).max() < 1e-7
    g = np.random.default_rng(0).normal(0, 1, 2 * n)
    gs, ga = ow.backprop(g)
    assert np.abs(J.T @ g - gs).max() < 1e-12
    # below the bounce threshold: inelastic, identity position Jacobians
    s2 = s.copy(); s2[n + 4] = -0.1
    ow2 = OracleWorld(md); nx2 = ow2.step(s2, a)
    assert abs(nx2[n + 4]) < 1e-9 and np.allclose(np.diag(ow2.getStateJacobian()[:n, :n]), 1.0)


def test_penetration_correction_adds_the_capped_velocity_to_the_normal_rows():
    """ContactConstraint.cpp:393-415 with the defaults of :45-47 (allowance 0, ERP 0.01, max ERV 1e-3): off by default; when the
    world enables it b_normal grows by min(depth * 0.01 / dt, 1e-3) and nothing else of the problem changes."""
    from oracle import OracleWorld
    from util import box_stack_inputs
    md, s, a = box_stack_inputs(4, 3)
    ow = OracleWorld(md)
    md.penetration_correction = True
    ow2 = OracleWorld(md)
    r0 = ow.step_batch(s, a, np.zeros_like(s), threads=1)
    r1 = ow2.step_batch(s, a, np.zeros_like(s), threads=1)
    assert (r0["status"] & 1).all()
    # the cubes separate faster with the correction on: by at most 1e-3 m/s * dt per step and contact
    dv = np.abs(r1["next"] - r0["next"]).max(1)
    assert dv.min() > 0 and dv.max() < 5e-3


def test_constrained_groups_are_solved_independently():
    """ConstraintSolver::buildConstrainedGroups (:724-780): two cubes side by side on the (immobile) ground are two constrained groups,
    each solved by its own run of the cascade - so a cube's step must not depend on the other cube being there: the two-cube
    world against the same world with the other cube parked 5 m up, bit for bit, including the worlds where one of the two needs the
    fallback stages."""
    import nimblephysics_amd as na
    from oracle import OracleWorld
    md = na.box_stack(); n = md.num_dofs; B = 400
    rng = np.random.default_rng(21)
    gb = md.boxes[0]
    top = (md.bodies[0].T_pj @ gb.T)[1, 3] + 0.5 * gb.size[1]
    half = 0.5 * gb.size[0]
    s = np.zeros((B, 2 * n))
    for k, x0 in enumerate((-0.4, 0.4)):
        o = 6 * k
        c0 = md.bodies[1 + k].T_pj[:3, 3]
        s[:, o + 1] = rng.uniform(-1.0, 1.0, B)
        over = rng.random(B) < 0.3
        s[:, o + 3] = np.where(over, np.sign(x0) * half * rng.uniform(0.93, 0.99, B), x0 * half + rng.uniform(-0.15, 0.15, B) * half) - c0[0]
        s[:, o + 4] = top + 0.1 - rng.uniform(1e-4, 1e-3, B) - c0[1]
        s[:, o + 5] = rng.uniform(-0.5, 0.5, B) * half - c0[2]
        s[:, n + o:n + o + 6] = rng.normal(0, 0.05, (B, 6))
    a = rng.normal(0, 0.1, (B, n))
    ow = OracleWorld(md)
    both = ow.step_batch(s, a, None, threads=4)
    mixed = 0
    for k in (0, 1):
        s1 = s.copy(); s1[:, 6 * (1 - k) + 4] += 5.0
        alone = ow.step_batch(s1, a, None, threads=4)
        ok = ((both["status"] | alone["status"]) & 0x80) == 0
        sl = [i for i in range(2 * n) if (i % n) // 6 == k]
        assert np.array_equal(both["next"][ok][:, sl], alone["next"][ok][:, sl])
        mixed += int((((alone["status"] & 0x2) != 0) & ((both["status"] & 0x2) == 0) & ok).sum())
    assert mixed > 10        # worlds where this cube resolves at stage 0 while the other one sends the world through the cascade


def _ref_boxbox():
    import os
    import oracle
    path = os.path.join(os.path.dirname(oracle.__file__), "_ref", "libdboxbox_ref.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.ref_collide_box_box.restype = C.c_int
    return lib


@pytest.mark.skipif(_ref_boxbox() is None, reason="oracle/_ref/libdboxbox_ref.so not built (needs the reference tree once: oracle/ref_build.py)")
def test_box_box_equals_the_references_own_dboxbox_on_random_pairs():
    """The reference's OWN box-box narrow phase (DARTCollide.cpp: dBoxBox + helpers + collideBoxBox, compiled from the reference's file
    by oracle/ref_build.py into oracle/_ref/libdboxbox_ref.so) against the oracle's restatement on thousands of random box pairs -
    touching face to face, edge to edge, corner to face, deep, barely, separated, aligned and tilted: same number of contacts in the
    same order, same contact types, points / normals / depths and the edge metadata bit for bit."""
    ref = _ref_boxbox()
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(2024)
    n_pairs = n_contacts = 0
    kinds = {}
    for trial in range(6000):
        s1 = rng.uniform(0.1, 1.0, 3); s2 = rng.uniform(0.1, 1.0, 3)
        mode = trial % 6
        if mode == 0:      # random orientations, centres at about touching distance
            R1 = Rotation.random(random_state=rng.integers(1 << 31)).as_matrix(); R2 = Rotation.random(random_state=rng.integers(1 << 31)).as_matrix()
            d = rng.normal(0, 1, 3); d /= np.linalg.norm(d)
            p2 = d * rng.uniform(0.2, 0.9) * 0.5 * (np.linalg.norm(s1) + np.linalg.norm(s2))
        elif mode in (1, 2):  # box 2 resting on top of box 1, yawed, slightly penetrating (face-face: 4..8 clipped points)
            R1 = np.eye(3); R2 = Rotation.from_euler("y", rng.uniform(-np.pi, np.pi)).as_matrix()
            if mode == 2: R2 = R2 @ Rotation.from_rotvec(rng.normal(0, 0.02, 3)).as_matrix()
            p2 = np.array([rng.uniform(-0.5, 0.5) * s1[0], 0.5 * (s1[1] + s2[1]) - rng.uniform(1e-4, 2e-2), rng.uniform(-0.5, 0.5) * s1[2]])
        elif mode == 3:    # edge over edge
            R1 = np.eye(3); R2 = Rotation.from_euler("zx", [np.pi / 4 + rng.normal(0, 0.1), rng.uniform(0.3, 1.2)]).as_matrix()
            p2 = np.array([rng.uniform(-0.3, 0.3) * s1[0], 0.5 * s1[1] + 0.5 * np.hypot(s2[0], s2[1]) - rng.uniform(1e-3, 2e-2), rng.uniform(-0.3, 0.3) * s1[2]])
        elif mode == 4:    # corner into a face
            R1 = np.eye(3); R2 = Rotation.from_euler("xz", [np.arctan(np.sqrt(2)) + rng.normal(0, 0.1), np.pi / 4 + rng.normal(0, 0.1)]).as_matrix()
            p2 = np.array([rng.uniform(-0.2, 0.2), 0.5 * s1[1] + 0.5 * np.linalg.norm(s2) - rng.uniform(1e-3, 3e-2), rng.uniform(-0.2, 0.2)])
        else:              # axis aligned, overlapping by a little along one axis
            R1 = np.eye(3); R2 = np.eye(3)
            ax = trial % 3
            p2 = rng.uniform(-0.3, 0.3, 3) * 0.5 * (s1 + s2); p2[ax] = 0.5 * (s1[ax] + s2[ax]) - rng.uniform(-1e-3, 2e-2)
        p1 = rng.normal(0, 1, 3)
        T1 = np.concatenate([R1.reshape(9), p1]); T2 = np.concatenate([R2.reshape(9), p1 + R1 @ p2 if mode == 0 else p1 + p2])
        o = np.zeros(16 * 22); r = np.zeros(16 * 22)
        no = L.nbo_box_box(_p(_d(T1)), _p(_d(s1)), _p(_d(T2)), _p(_d(s2)), C.c_double(0.03), _p(o))
        nr = ref.ref_collide_box_box(_p(_d(s1)), _p(_d(T1)), _p(_d(s2)), _p(_d(T2)), C.c_double(0.03), _p(r), 16)
        assert no == nr, (trial, mode, no, nr)
        if nr:
            n_pairs += 1; n_contacts += nr
            a, b = o[:22 * min(nr, 8)].reshape(-1, 22), r[:22 * min(nr, 8)].reshape(-1, 22)
            assert np.array_equal(a[:, 7], b[:, 7]), (trial, mode, a[:, 7], b[:, 7])           # contact types
            assert np.array_equal(a[:, :7], b[:, :7]), (trial, mode, np.abs(a[:, :7] - b[:, :7]).max())   # point, normal, depth: bit for bit
            edge = a[:, 7] == 3
            # edge fixed points and directions (unit vectors: Eigen's normalized() divides by the norm, and so do the stand-in and the oracle)
            assert np.abs(a[edge, 8:20] - b[edge, 8:20]).max(initial=0.0) <= 0.0, (trial, mode, np.abs(a[edge, 8:20] - b[edge, 8:20]).max(0))
            for t in a[:, 7]: kinds[int(t)] = kinds.get(int(t), 0) + 1
    assert n_pairs > 3000 and n_contacts > 8000 and set(kinds) == {1, 2, 3}, (n_pairs, n_contacts, kinds)


@pytest.mark.skipif(_ref_boxbox() is None, reason="oracle/_ref/libdboxbox_ref.so not built (needs the reference tree once: oracle/ref_build.py)")
def test_sphere_narrow_phases_equal_the_references_own_functions_on_random_pairs():
    """collideBoxSphere / collideSphereBox / collideSphereSphere of the reference (DARTCollide.cpp:1482-1882, compiled from the reference's
    file into oracle/_ref/libdboxbox_ref.so) against the oracle's restatement: sphere centre outside the box (face, edge and corner
    regions: one, two or three locked faces), inside the box, touching, too deep, separated; sphere pairs.  Same count and type,
    same locked faces and face normals; point, normal, depth and the sphere data to 1e-15 (the box-frame transforms go through Eigen's
    products in the reference, whose summation order the stand-in follows but cannot prove)."""
    ref = _ref_boxbox()
    ref.ref_collide_sphere.restype = C.c_int
    L.nbo_sphere_pair.restype = C.c_int
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(77)
    seen = {}
    for trial in range(9000):
        which = trial % 3
        Rb = Rotation.random(random_state=rng.integers(1 << 31)).as_matrix()
        pb = rng.normal(0, 1, 3)
        half = rng.uniform(0.1, 0.8, 3)
        rad = rng.uniform(0.05, 0.5)
        if which < 2:
            region = trial % 7                                       # how many coordinates of the centre lie outside the box
            loc = rng.uniform(-1, 1, 3) * half
            out_axes = rng.permutation(3)[: (region % 4)]
            for ax in out_axes:
                loc[ax] = np.sign(rng.normal()) * (half[ax] + rad * rng.uniform(0.3, 1.05) / np.sqrt(max(len(out_axes), 1)))
            if region == 6:                                         # centre just inside a face, small sphere (deeper ones exceed the clipping depth)
                loc = rng.uniform(-0.9, 0.9, 3) * half; ax = int(rng.integers(3))
                rad = rng.uniform(0.002, 0.012); loc[ax] = np.sign(rng.normal()) * (half[ax] - rng.uniform(0.001, 0.012))
            cs = pb + Rb @ loc
            Tb = np.concatenate([Rb.reshape(9), pb]); Ts = np.concatenate([np.eye(3).reshape(9), cs])
            sb = 2 * half; ss = np.array([rad, rad, rad])
            args = (Tb, sb, Ts, ss) if which == 0 else (Ts, ss, Tb, sb)
        else:
            r2 = rng.uniform(0.05, 0.5)
            d = rng.normal(0, 1, 3); d /= np.linalg.norm(d)
            c2 = pb + d * (rad + r2) * rng.uniform(0.9, 1.03)
            args = (np.concatenate([np.eye(3).reshape(9), pb]), np.array([rad] * 3), np.concatenate([np.eye(3).reshape(9), c2]), np.array([r2] * 3))
        o = np.zeros(4 * 32); r = np.zeros(4 * 32)
        no = L.nbo_sphere_pair(which, _p(_d(args[0])), _p(_d(args[1])), _p(_d(args[2])), _p(_d(args[3])), C.c_double(0.03), _p(o))
        nr = ref.ref_collide_sphere(which, _p(_d(args[1])), _p(_d(args[0])), _p(_d(args[3])), _p(_d(args[2])), C.c_double(0.03), _p(r), 4)
        assert no == nr, (trial, which, no, nr)
        if nr:
            a, b = o[:32], r[:32]
            assert a[7] == b[7] and np.array_equal(a[20:23], b[20:23]), (trial, which, a[7], b[7], a[20:23], b[20:23])   # type, locked faces
            assert np.abs(a - b).max() <= 1e-15 * max(1.0, np.abs(b).max()), (trial, which, np.abs(a - b).max())
            seen[(which, int(b[7]), int(b[20:23].sum()))] = seen.get((which, int(b[7]), int(b[20:23].sum())), 0) + 1
    # box-sphere and sphere-box with 1, 2, 3 locked faces, the centre-inside cases (plain vertex-face types), sphere-sphere
    for key in [(0, 5, 1), (0, 5, 2), (0, 5, 3), (1, 4, 1), (1, 4, 2), (1, 4, 3), (0, 2, 0), (1, 1, 0), (2, 6, 0)]:
        assert seen.get(key, 0) > 5, (key, seen)


def test_the_soak_instruments_leave_the_oracle_alone_when_off_and_replay_its_own_solution():
    """OracleWorld.set_lcp_noise / set_lcp_forced / set_pinv_noise are instruments of the randomised soaks (tools/soak_parity.py), not reference behaviour:
    switched off they change nothing; one-ulp noise on A leaves a well-posed world's answer within 1e-9; the oracle's OWN cascade
    solution forced back in as the solver's output reproduces next state and gradients bit for bit; a solution that violates the LCP is
    refused (0x40000000)."""
    md, s, a = contact_inputs("atlas20", 64, 7, joint_noise=0.05)
    g = np.random.default_rng(0).normal(0, 1, s.shape)
    w = OracleWorld(md)
    ref = w.step_batch(s, a, g, threads=2)
    cascade = [i for i in range(len(s)) if (ref["status"][i] & 0xC) and not (ref["status"][i] & 0x10)]
    assert cascade, "no world of the sample resolves at the pivoting or the CFM + PGS stage"
    w.set_lcp_noise(1, 5)
    noisy = w.step_batch(s, a, g, threads=2)
    w.set_lcp_noise(0)
    again = w.step_batch(s, a, g, threads=2)
    for k in ("next", "grad_state", "grad_action"):
        assert np.array_equal(again[k], ref[k])
    stage0 = (ref["status"] & 0x2) != 0
    assert stage0.any() and np.abs(noisy["next"][stage0] - ref["next"][stage0]).max() < 1e-9
    i = cascade[0]
    w.reset_lcp_cache(); nx = w.step(s[i], a[i]); x = w.last_lcp()["x"].copy(); gs, ga = w.backprop(g[i])
    pgs = bool(ref["status"][i] & 0x8)
    w.reset_lcp_cache(); w.set_lcp_forced(x, cfm_stage=pgs)
    nx2 = w.step(s[i], a[i]); st2 = w.last_status; gs2, ga2 = w.backprop(g[i])
    assert not (st2 & 0x40000000) and (st2 & (0x8 if pgs else 0x4))
    assert np.array_equal(nx, nx2) and np.array_equal(gs, gs2) and np.array_equal(ga, ga2)
    w.reset_lcp_cache(); w.set_lcp_forced(-np.abs(x) - 1.0)
    w.step(s[i], a[i])
    assert w.last_status & 0x40000000
    w.set_lcp_forced(None)
    w.reset_lcp_cache()
    assert np.array_equal(w.step(s[i], a[i]), nx)
    # the fourth instrument (round 5): one ulp on every entry of the BACKWARD pass's pseudo-inverse.  A well-conditioned world does not notice
    # (the gradients of the stage-0 worlds move by < 1e-9, the next state not at all); switched off it changes nothing
    w.set_pinv_noise(1, 3)
    noisy = w.step_batch(s, a, g, threads=2)
    w.set_pinv_noise(0)
    assert np.array_equal(noisy["next"], ref["next"])
    scale = np.abs(ref["grad_state"]).max()
    assert np.abs(noisy["grad_state"][stage0] - ref["grad_state"][stage0]).max() < 1e-9 * scale
    again = w.step_batch(s, a, g, threads=2)
    for k in ("next", "grad_state", "grad_action"):
        assert np.array_equal(again[k], ref[k])


def test_the_oracle_flags_a_contact_kept_after_sixteen_distinct_narrow_phase_points():
    """The device's duplicate filter remembers 16 distinct points per world (kept or dropped by the depth filter); a contact kept after
    that may be an unnoticed duplicate, the device flags the world (NBL_ST_CONTACT_OVERFLOW) and the oracle raises the same flag by
    the same rule, so that a comparison can leave the world out (tests/test_gpu_contact.py has the device side)."""
    from util import duplicate_filter_scene
    for deep, flagged in ((4, True), (3, False)):
        md, s, a = duplicate_filter_scene(deep)
        w = OracleWorld(md)
        w.step(s[0], a[0])
        assert len(w.last_contacts()) == 4 and (w.last_status & 0x1)
        assert bool(w.last_status & 0x80) == flagged, (deep, hex(w.last_status))
