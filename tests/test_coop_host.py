"""The PRODUCT's wave-cooperative LCP device code (nimblephysics_amd/csrc/coop_dev.hpp: one world per wavefront, lane =
LCP row / matrix column) compiled for the host on a thread-per-lane wave emulation (tests/host_shim/wave_emu.hpp) and
checked against
  * numpy's pseudo-inverse (the role of Eigen's completeOrthogonalDecomposition in CGGM.cpp:280 / LCPUtils.cpp:113),
  * the one-world-per-lane statement of the same stage-0 algorithm (tests/host_shim/lane_lcp_statement.hpp: guessSolution +
    the constructMatrices / opportunisticallyStandardizeResults loop, plain sequential code).
The emulation deadlocks if lanes disagree on control flow around a cross-lane primitive, so passing also shows that the
device code is wave-uniform where it has to be.  A checker for device code, not a CPU path of the product."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)


def _build_shim(maxc):
    src = os.path.join(HERE, "host_shim", "coop_shim.cpp")
    out = os.path.join(HERE, "host_shim", "libcoop_shim.so" if maxc == 8 else f"libcoop_shim{maxc}.so")
    deps = [src, os.path.join(HERE, "host_shim", "wave_emu.hpp")] + \
        [os.path.join(ROOT, "nimblephysics_amd", "csrc", f) for f in ("coop_dev.hpp", "coop_dantzig_dev.hpp", "lcp_dev.hpp", "spatial_dev.hpp")] + [os.path.join(HERE, "host_shim", "lane_lcp_statement.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-pthread", f"-DNBL_MAXC={maxc}", "-I", os.path.join(HERE, "host_shim"),
                               "-I", os.path.join(ROOT, "nimblephysics_amd", "csrc"), "-o", out, src])
    lib = C.CDLL(out)
    lib.R = lib.shim_rows()          # LCP rows of this instantiation of the device code (24 / 48)
    lib.NC = lib.R // 3
    assert lib.R == 3 * maxc
    return lib


@pytest.fixture(scope="module", params=[8, 16], ids=["rows24", "rows48"])
def shim(request):
    """The device code of BOTH instantiations of the library (csrc/abi_variants.h): 8 contacts / 24 LCP rows and 16 contacts / 48 rows."""
    return _build_shim(request.param)


@pytest.fixture(scope="module")
def shim24():
    return _build_shim(8)


def _p(a):
    return a.ctypes.data_as(pd)


def _pi(a):
    return a.ctypes.data_as(pi)


def test_coop_pinv_equals_numpy_pinv_on_masked_rank_deficient_systems(shim):
    """Full-rank and rank-deficient, symmetric and non-symmetric, with rows/columns masked out (zero) the way the
    kernels select the clamping block: Q^+ to 1e-9, rank exact."""
    R, NC = shim.R, shim.NC
    rng = np.random.default_rng(0)
    for trial in range(50 if R == 24 else 24):
        c = int(rng.integers(1, R + 1)); k = int(rng.integers(1, c + 1))
        idx = np.sort(rng.choice(R, c, replace=False))
        U = rng.normal(0, 1, (c, k)); V = rng.normal(0, 1, (c, k))
        sub = U @ U.T if trial % 2 == 0 else U @ V.T
        Q = np.zeros((R, R)); Q[np.ix_(idx, idx)] = sub
        P = np.zeros((R, R))
        rank = shim.shim_coop_pinv(_p(np.ascontiguousarray(Q)), c, _p(P))
        ref = np.linalg.pinv(Q, rcond=1e-11)
        assert rank == k
        assert np.abs(P - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-30)
    Z = np.zeros((R, R)); P = np.ones((R, R))
    assert shim.shim_coop_pinv(_p(Z), 5, _p(P)) == 0 and not P.any()


def test_coop_pinv_on_contact_matrices_with_friction_rows_on_their_bound(shim):
    """The matrices the Householder route is there for: Q = A(c, c) + A(c, u) E of a robot standing on two flat feet, some friction rows on
    their bound folded into their normal's column - non-symmetric, up to 23 rows of rank 12, the body inertias spread over 1e2.  The
    rank-deficient completion (24-row build: R = R1 [I W], Cholesky of I + W W^T; 48-row build: the second Householder pass) against
    numpy's pseudo-inverse: rank exact, Q^+ to cond(Q) eps."""
    R, NC = shim.R, shim.NC
    rng = np.random.default_rng(11)
    worst, tested = 0.0, 0
    for trial in range(60 if R == 24 else 16):
        nb = int(rng.integers(1, 3))                                   # one or two 6-DOF bodies
        J = np.zeros((R, 6 * nb))
        for cidx in range(NC):                                          # contact -> body, point on its sole
            bdy = int(rng.integers(0, nb)); pt = np.array([rng.uniform(-0.1, 0.1), 0.0, rng.uniform(-0.05, 0.05)])
            for ax in range(3):
                d = np.eye(3)[[1, 0, 2][ax]]
                J[3 * cidx + ax, 6 * bdy:6 * bdy + 3] = np.cross(pt, d); J[3 * cidx + ax, 6 * bdy + 3:6 * bdy + 6] = d
        Lam = np.zeros((6 * nb, 6 * nb))
        for bdy in range(nb):
            M = rng.normal(0, 1, (6, 6)); Lam[6 * bdy:6 * bdy + 6, 6 * bdy:6 * bdy + 6] = M @ np.diag(10 ** rng.uniform(-1, 1, 6)) @ M.T
        A = J @ Lam @ J.T
        cls = rng.choice([0, 1, 2], size=R, p=[0.15, 0.6, 0.25])      # not clamping / clamping / upper bound (friction rows only)
        cls[0::3] = np.where(cls[0::3] == 2, 1, cls[0::3])
        Q = np.zeros((R, R))
        cl = np.where(cls == 1)[0]
        if len(cl) == 0:
            continue
        Q[np.ix_(cl, cl)] = A[np.ix_(cl, cl)]
        for u in np.where(cls == 2)[0]:
            nrm = u - u % 3
            if cls[nrm] == 1:
                Q[cl, nrm] += rng.choice([-1.0, 1.0]) * rng.uniform(0.3, 1.0) * A[cl, u]
        sub = Q[np.ix_(cl, cl)]
        sv = np.linalg.svd(sub, compute_uv=False)
        k = int((sv > 1e-12 * sv[0]).sum())
        cond = sv[0] / sv[k - 1]
        if cond > 1e8:                                                  # (no clear gap between the singular values that count and round-off)
            continue
        tested += 1
        P = np.zeros((R, R))
        rank = shim.shim_coop_pinv(_p(np.ascontiguousarray(Q)), len(cl), _p(P))
        assert rank == k, (trial, rank, k, len(cl))
        ref = np.linalg.pinv(Q, rcond=0.5 * sv[k - 1] / sv[0])
        e = np.abs(P - ref).max() / np.abs(ref).max()
        worst = max(worst, e / (cond * 2.2e-16))
        assert e <= 500 * cond * 2.2e-16, (trial, e, cond, k, len(cl))
    assert tested >= (40 if R == 24 else 10), tested
    print("coopPinv on contact matrices: worst error in units of cond(Q) eps:", worst, "over", tested, "matrices")


def test_coop_pinv_sym_equals_numpy_pinv_on_positive_semidefinite_systems(shim):
    """coopPinvSym: the pseudo-inverse of symmetric positive semi-definite Q by two Cholesky factorisations, Q = G G^T (diagonally
    pivoted, rank by the threshold of the reference's complete orthogonal decomposition) and Q^+ = G (G^T G)^-2 G^T - the route the
    kernels take whenever no friction row sits on its bound.  Random masked rank-deficient and full-rank blocks, contact matrices
    J D J^T of two flat feet (rank 12 of 24), the same with the fallback CFM on the diagonal (full rank, cond ~ 1e5): Q^+ and the
    rank against numpy, and against the Householder route (coopPinv) on the same input."""
    R, NC = shim.R, shim.NC
    rng = np.random.default_rng(7)
    worst = 0.0
    for trial in range(80 if R == 24 else 32):
        c = int(rng.integers(1, R + 1)); idx = np.sort(rng.choice(R, c, replace=False))
        if trial % 4 == 3:                         # a standing robot: c rows on two 6-DOF bodies (+ the CFM every other time)
            k = min(c, 12)
            J = rng.normal(0, 1, (c, 12)); sub = J @ np.diag(rng.uniform(0.1, 2, 12)) @ J.T
            if trial % 8 == 7:
                sub = sub + 1e-4 * np.eye(c); k = c
        else:
            k = int(rng.integers(1, c + 1))
            U = rng.normal(0, 1, (c, k)); sub = U @ U.T
        Q = np.zeros((R, R)); Q[np.ix_(idx, idx)] = sub
        Q = 0.5 * (Q + Q.T)
        P = np.zeros((R, R)); P2 = np.zeros((R, R))
        rank = shim.shim_coop_pinv_sym(_p(np.ascontiguousarray(Q)), c, _p(P))
        rank2 = shim.shim_coop_pinv(_p(np.ascontiguousarray(Q)), c, _p(P2))
        assert rank == k and rank2 == k, (trial, rank, rank2, k)
        sv = np.linalg.svd(sub, compute_uv=False)
        cond = sv[0] / sv[k - 1]
        ref = np.linalg.pinv(Q, rcond=0.5 * sv[k - 1] / sv[0])
        scale = max(np.abs(ref).max(), 1e-30)
        e = np.abs(P - ref).max() / scale
        worst = max(worst, e / (cond * 2.2e-16))
        assert e <= 200 * cond * 2.2e-16, (trial, e, cond)               # cond(Q) eps, like the QR route
        assert np.abs(P - P2).max() / scale <= 400 * cond * 2.2e-16
        assert np.abs(P - P.T).max() <= 1e-12 * scale
    print("coopPinvSym: worst error in units of cond(Q) eps:", worst)
    # an exactly singular contact matrix whose last Cholesky pivot (round-off) lands above eps * size: rank 5 like the COD's
    import json
    f = json.load(open(os.path.join(HERE, "golden", "pinv_rank_borderline.json")))
    A6 = np.array([[float.fromhex(x) for x in row] for row in f["A"]])
    Q = np.zeros((R, R)); Q[:6, :6] = A6
    P = np.zeros((R, R)); P2 = np.zeros((R, R))
    assert shim.shim_coop_pinv_sym(_p(np.ascontiguousarray(Q)), 6, _p(P)) == 5 and shim.shim_coop_pinv(_p(np.ascontiguousarray(Q)), 6, _p(P2)) == 5
    assert np.abs(P - P2).max() <= 1e-10 * np.abs(P2).max()
    Z = np.zeros((R, R)); P = np.ones((R, R))
    assert shim.shim_coop_pinv_sym(_p(Z), 5, _p(P)) == 0 and not P.any()


def _contact_problem(rng, trial, R=24):
    NC = R // 3
    nc = int(rng.integers(1, NC + 1)); m = 3 * nc
    ndof = int(rng.choice([6, 12, 30]))
    J = rng.normal(0, 1, (m, ndof))
    A = np.zeros((R, R)); A[:m, :m] = J @ np.diag(rng.uniform(0.1, 2, ndof)) @ J.T
    mu = np.zeros(NC); mu[:nc] = rng.choice([0.5, 1.0], nc)
    xs = np.zeros(R)
    for c in range(nc):
        if rng.random() < 0.7:
            xs[3 * c] = rng.uniform(0.1, 2)
            xs[3 * c + 1:3 * c + 3] = rng.uniform(-0.5, 0.5, 2) * mu[c] * xs[3 * c]
            if rng.random() < 0.4:       # sliding: one friction row on its bound
                xs[3 * c + 1 + int(rng.integers(0, 2))] = rng.choice([-1, 1]) * mu[c] * xs[3 * c]
    b = np.zeros(R); b[:m] = (A @ xs)[:m]
    kind = trial % 3
    if kind == 1:
        b[:m] += rng.normal(0, 0.05, m)
    if kind == 2:
        b[:m] = rng.normal(0, 1, m)
    have = int(trial % 5 >= 3)
    xc = np.zeros(R); xc[:m] = xs[:m] + rng.normal(0, 1e-3, m) * (trial % 2)
    return m, A, b, mu, have, xc


def test_coop_stage0_equals_the_one_world_per_lane_statement(shim):
    """guessSolution / warm start -> classification -> least-squares standardisation -> isLCPSolutionValid, on resting,
    sliding (upper-bound rows), perturbed and random contact problems of 1..8 contacts with rank-deficient A: same
    accept/reject decision, same x, same classes, and the pseudo-inverse left in LDS is that of the final Q."""
    R, NC = shim.R, shim.NC
    rng = np.random.default_rng(1)
    n_ok = n_ub = n_fail = 0
    for trial in range(240 if R == 24 else 90):
        m, A, b, mu, have, xc = _contact_problem(rng, trial, R)
        X1 = np.zeros(R); X01 = np.zeros(R); c1 = np.zeros(R, np.int32); E1 = np.zeros(R); P = np.zeros((R, R))
        X2 = np.zeros(R); X02 = np.zeros(R); c2 = np.zeros(R, np.int32); E2 = np.zeros(R)
        r1 = shim.shim_coop_stage0(m, _p(A), _p(b), _p(mu), have, _p(xc), _p(X1), _p(X01), _pi(c1), _p(E1), _p(P))
        r2 = shim.shim_lane_stage0(m, _p(A), _p(b), _p(mu), have, _p(xc), _p(X2), _p(X02), _pi(c2), _p(E2))
        assert (r1 & 1) == r2
        assert np.abs(X01 - X02).max() <= 1e-9 * max(1.0, np.abs(X02).max())          # pre-solve x handed to the cascade
        if not r2:
            n_fail += 1
            continue
        n_ok += 1
        assert np.abs(X1 - X2).max() <= 1e-9 * max(1.0, np.abs(X2).max())
        assert np.array_equal(c1[:m], c2[:m]) and np.array_equal(E1, E2)
        n_ub += int((c1 == 2).any())
        if (r1 & 2):
            cl = c1 == 1
            Q = np.zeros((R, R)); Q[np.ix_(cl, cl)] = A[np.ix_(cl, cl)]
            for u in np.where(c1 == 2)[0]:          # upper-bound rows ride on their normal column
                Q[cl, u - u % 3] += E1[u] * A[cl, u]
            ref = np.linalg.pinv(Q, rcond=1e-11)
            assert np.abs(P - ref).max() <= 1e-7 * max(np.abs(ref).max(), 1e-30)
    assert n_ok > (50 if R == 24 else 12) and n_ub > (10 if R == 24 else 3) and n_fail > (50 if R == 24 else 12)


# ---- stages 1-3 of the solver cascade, cooperative (coop_dantzig_dev.hpp) ----
import oracle  # noqa: E402
from util import contact_lcp, have_ref  # noqa: E402

OL = oracle._lib()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_coop_dantzig_is_bit_identical_to_the_reference_dsolvelcp(shim):
    """The wave-cooperative Dantzig driver restates dSolveLCP operation by operation (the reference's L / d factor in its own
    row order, dLDLTAddTL / dLDLTRemove down-dates, the blocked summation order of dSolveL1 / dSolveL1T, dDot's running sum, no
    fused multiply-adds): against the reference's own solver (oracle/_ref) the success flag and EVERY BIT of x must be equal,
    on full-rank problems and on rank-deficient ones (6- and 3-DOF bodies with up to 8 frictional contacts) where A(C,C)
    goes singular and the s <= 0 early exit is decided by round-off."""
    R, NC = shim.R, shim.NC
    rng = np.random.default_rng(0)
    solved = failed = 0
    for trial in range(40 if R == 24 else 14):
        nc = int(rng.integers(1, NC + 1)); n = 3 * nc
        ndof = n + int(rng.integers(0, 6)) if trial % 3 == 0 else int(rng.choice([3, 6, 12]))   # 2 of 3: rank-deficient A
        A, b, lo, hi, fi = contact_lcp(rng, nc, ndof)
        xr = np.zeros(n); xd = np.zeros(n)
        okr = OL.nbo_lcp_dantzig(n, _p(A), _p(xr), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), _pi(fi.copy()), 1)
        okd = shim.shim_coop_dantzig(n, _p(A), _p(xd), _p(b), _p(lo), _p(hi), _pi(fi))
        if okd == -1:                      # NaN step: the reference carries the NaN into x (its caller then resets x)
            assert okr == 0 or not np.all(np.isfinite(xr)), (trial, okr)
            continue
        assert okr == okd, (trial, n, ndof, okr, okd)
        if okr == 1:
            solved += 1
            assert np.array_equal(xr, xd), (trial, n, ndof, np.abs(xr - xd).max())
        else:
            failed += 1
    assert solved > (20 if R == 24 else 4) and failed > 0


def test_coop_pgs_and_reduce_equal_the_oracle_restatement(shim):
    R, NC = shim.R, shim.NC
    rng = np.random.default_rng(2)
    for trial in range(21 if R == 24 else 6):
        nc = int(rng.integers(1, NC + 1)); n = 3 * nc
        A, b, lo, hi, fi = contact_lcp(rng, nc, int(rng.integers(3, R)), cfm=1e-4)
        if trial % 3 == 0 and nc >= 2:      # duplicate a contact so that reduce() has something to merge
            A[3:6, :] = A[0:3, :]; A[:, 3:6] = A[:, 0:3]; b[3:6] = b[0:3]; hi[3:6] = hi[0:3]; lo[3:6] = lo[0:3]
        x0 = rng.normal(0, 0.1, n)
        xo = x0.copy(); xd = x0.copy()
        oko = OL.nbo_lcp_pgs(n, _p(A), _p(xo), _p(b), _p(lo), _p(hi), _pi(fi), 30, C.c_double(1e-6), C.c_double(1e-3), C.c_double(1e-9))
        okd = shim.shim_coop_pgs(n, _p(A), _p(xd), _p(b), _p(lo), _p(hi), _pi(fi))
        assert oko == okd and np.allclose(xo, xd, rtol=1e-12, atol=1e-14)
        for rf in (0, 1):
            Ar = np.zeros(n * n); xr = np.zeros(n); br = np.zeros(n); lor = np.zeros(n); hir = np.zeros(n); fr = np.zeros(n, np.int32); mo = np.zeros(n * n)
            nr = OL.nbo_lcp_reduce(n, _p(A), _p(x0), _p(b), _p(lo), _p(hi), _pi(fi), rf, _p(Ar), _p(xr), _p(br), _p(lor), _p(hir), _pi(fr), _p(mo))
            Ad = np.zeros(n * n); xdv = np.zeros(n); bd = np.zeros(n); lod = np.zeros(n); hid = np.zeros(n); fd = np.zeros(n, np.int32); mt = np.zeros(n, np.int32)
            nd = shim.shim_coop_reduce(n, _p(A), _p(x0), _p(b), _p(lo), _p(hi), _pi(fi), rf, _p(Ad), _p(xdv), _p(bd), _p(lod), _p(hid), _pi(fd), _pi(mt))
            assert nr == nd
            assert np.allclose(Ar[:nr * nr], Ad[:nr * nr]) and np.allclose(br[:nr], bd[:nr]) and np.array_equal(fr[:nr], fd[:nr])
            assert np.allclose(xr[:nr], xdv[:nr]) and np.allclose(lor[:nr], lod[:nr]) and np.allclose(hir[:nr], hid[:nr])
            M = mo[:n * nr].reshape(n, nr)
            for o in range(n):
                assert (M[o].sum() == 0 and mt[o] == -1) or (M[o, mt[o]] == 1 and M[o].sum() == 1)


def test_stage0_on_one_constrained_group_equals_stage0_of_that_group_alone(shim):
    """A world with two constrained groups is a block-diagonal LCP; the kernels run stage 0 group by group with the other group's
    rows switched off (CoopRow::on).  The result on a group's rows must be what stage 0 gives on that group's own problem, whatever
    position its rows have in the world (first or second group, 3 + 4 or 4 + 3 contacts)."""
    R, NC = shim.R, shim.NC
    rng = np.random.default_rng(11)
    for trial in range(40 if R == 24 else 10):
        nA, nB = int(rng.integers(1, NC // 2 + 1)), int(rng.integers(1, NC // 2 + 1))
        parts = []
        for nc in (nA, nB):
            m = 3 * nc
            ndof = int(rng.choice([6, 12]))
            J = rng.normal(0, 1, (m, ndof))
            A = J @ np.diag(rng.uniform(0.1, 2, ndof)) @ J.T
            xs = np.zeros(m)
            for c in range(nc):
                if rng.random() < 0.8:
                    xs[3 * c] = rng.uniform(0.1, 2); xs[3 * c + 1:3 * c + 3] = rng.uniform(-0.4, 0.4, 2) * xs[3 * c]
            b = A @ xs + (rng.normal(0, 0.05, m) if trial % 2 else 0.0)
            parts.append((m, A, b))
        mT = parts[0][0] + parts[1][0]
        A = np.zeros((R, R)); b = np.zeros(R); mu = np.ones(NC)
        A[:parts[0][0], :parts[0][0]] = parts[0][1]; A[parts[0][0]:mT, parts[0][0]:mT] = parts[1][1]
        b[:parts[0][0]] = parts[0][2]; b[parts[0][0]:mT] = parts[1][2]
        off = 0
        for (m, Ag, bg) in parts:
            mask = ((1 << m) - 1) << off
            X = np.zeros(R); X0 = np.zeros(R); cls = np.zeros(R, np.int32); E = np.zeros(R)
            ret = shim.shim_coop_stage0_masked(mT, _p(np.ascontiguousarray(A)), _p(b), _p(mu), C.c_uint64(mask), _p(X), _p(X0), _pi(cls), _p(E))
            A1 = np.zeros((R, R)); A1[:m, :m] = Ag; b1 = np.zeros(R); b1[:m] = bg
            X1 = np.zeros(R); X01 = np.zeros(R); cls1 = np.zeros(R, np.int32); E1 = np.zeros(R); P1 = np.zeros((R, R))
            ret1 = shim.shim_coop_stage0(m, _p(np.ascontiguousarray(A1)), _p(b1), _p(mu), 0, _p(np.zeros(R)), _p(X1), _p(X01), _pi(cls1), _p(E1), _p(P1))
            assert (ret & 1) == (ret1 & 1), (trial, off, ret, ret1)
            assert np.array_equal(cls[off:off + m], cls1[:m]), (trial, off)
            assert np.abs(X[off:off + m] - X1[:m]).max() <= 1e-10 * max(np.abs(X1).max(), 1e-30), (trial, off)
            assert np.abs(X0[off:off + m] - X01[:m]).max() <= 1e-10 * max(np.abs(X01).max(), 1e-30)
            assert not X[:off].any() and not X[off + m:].any()
            off += m


def test_cascade_on_one_constrained_group_equals_the_cascade_of_that_group_alone(shim):
    """The same for stages 1-3, the order of preference and the standardisation that follows (CFM on the diagonal included):
    problems that stage 0 cannot resolve (random b, rank-deficient A), as the second and as the first group of a two-group world."""
    R, NC = shim.R, shim.NC
    rng = np.random.default_rng(12)
    seen = set()
    for trial in range(7 if R == 24 else 2):
        parts = []
        for nc in (int(rng.integers(1, NC // 2 + 1)), int(rng.integers(1, NC // 2 + 1))):
            m = 3 * nc
            ndof = int(rng.choice([3, 6, 12]))
            J = rng.normal(0, 1, (m, ndof))
            A = J @ np.diag(rng.uniform(0.1, 2, ndof)) @ J.T
            b = rng.normal(0, 1, m) if trial % 3 else A @ np.abs(rng.normal(0, 1, m))
            parts.append((m, A, b, rng.normal(0, 0.1, m)))
        mT = parts[0][0] + parts[1][0]
        A = np.zeros((R, R)); b = np.zeros(R); x0 = np.zeros(R); mu = np.ones(NC)
        o = 0
        for (m, Ag, bg, xg) in parts:
            A[o:o + m, o:o + m] = Ag; b[o:o + m] = bg; x0[o:o + m] = xg; o += m
        off = 0
        for (m, Ag, bg, xg) in parts:
            mask = ((1 << m) - 1) << off
            X = np.zeros(R); Xs = np.zeros(R); cls = np.zeros(R, np.int32); cfm = C.c_double(0)
            st = shim.shim_coop_cascade_masked(mT, _p(np.ascontiguousarray(A)), _p(b), _p(mu), _p(x0), C.c_uint64(mask), C.c_double(1e-4), _p(X), C.byref(cfm), _p(Xs), _pi(cls))
            A1 = np.zeros((R, R)); A1[:m, :m] = Ag; b1 = np.zeros(R); b1[:m] = bg; x1 = np.zeros(R); x1[:m] = xg
            X1 = np.zeros(R); Xs1 = np.zeros(R); cls1 = np.zeros(R, np.int32); cfm1 = C.c_double(0)
            st1 = shim.shim_coop_cascade_masked(m, _p(np.ascontiguousarray(A1)), _p(b1), _p(mu), _p(x1), C.c_uint64((1 << m) - 1), C.c_double(1e-4), _p(X1), C.byref(cfm1), _p(Xs1), _pi(cls1))
            assert st == st1 and cfm.value == cfm1.value, (trial, off, hex(st), hex(st1))
            assert np.array_equal(X[off:off + m], X1[:m]), (trial, off)                  # the solvers see the same problem: bit for bit
            assert np.array_equal(cls[off:off + m], cls1[:m]) and np.abs(Xs[off:off + m] - Xs1[:m]).max() <= 1e-10 * max(np.abs(Xs1).max(), 1e-30)
            assert not X[:off].any() and not X[off + m:].any()
            seen.add(st & 0x13c)
            off += m
    assert len(seen) >= (3 if R == 24 else 2), seen      # the trials reach several exits of the cascade


def test_reverse_mode_of_the_so3_integration_is_exact_up_to_the_log_map_singularity(shim24):
    shim = shim24
    """spatial_dev.hpp so3IntegrationVjp (the VJP of q' = logMap(exp(q) exp(w dt)) the device uses where the reference finite-differences,
    BallJoint.cpp:351-408 / FreeJoint.cpp:950-1007) against a five-point stencil of the same function in 80-bit arithmetic, from a generic
    rotation down to 2e-3 rad short of pi, where logMap in doubles loses digits like 1e-16 / gap^2 (and a quotient in doubles with eps 1e-6,
    the reference's, is off by 1e-4)."""
    import sys
    sys.path.insert(0, HERE)
    from test_gpu_ball_joint import _so3_vjp_extended_precision
    rng = np.random.default_rng(21)
    dt = 1e-3
    for gap, tol in ((2.0, 1e-9), (0.5, 1e-9), (1e-1, 1e-9), (1e-2, 1e-7), (2e-3, 1e-6)):
        for _ in range(8):
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            q = ax * (np.pi - gap); w = rng.normal(0, 0.5, 3) * rng.choice([1.0, 10.0]); g = rng.normal(0, 1, 3)
            pT = np.zeros(3); vT = np.zeros(3)
            shim.shim_so3_integration_vjp(_p(q), _p(w), C.c_double(dt), _p(g), _p(pT), _p(vT))
            x = _so3_vjp_extended_precision(q, w, dt, g)
            assert np.abs(np.concatenate([pT, vT]) - x).max() <= tol * np.abs(x).max(), (gap, np.abs(np.concatenate([pT, vT]) - x).max())
