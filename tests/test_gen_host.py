"""The product's GENERAL-m contact LCP code (nimblephysics_amd/csrc/gen_lcp_dev.hpp, gen_dantzig_dev.hpp: any number of rows up to
192 = 64 contacts; what a model with max_contacts > 16 runs) compiled for the host under a one-lane wave policy
(tests/host_shim/gen_shim.cpp), against
  * numpy's pseudo-inverse on masked, rank-deficient, symmetric and non-symmetric systems of up to 192 rows,
  * the REFERENCE'S OWN dSolveLCP (oracle/_ref/libodelcp_ref.so): success flag and every bit of x, up to 192 rows, rank-deficient
    contact problems included,
  * the wavefront-cooperative device code of the 24- and 48-row builds (tests/host_shim/coop_shim.cpp) on the problems both can hold:
    same stage-0 decision, x, row classes; same cascade stage and solution,
  * the oracle's solver cascade (stage 1 = the reference's own Dantzig) on problems of 20 - 64 contacts and on the reference's fixtures.
A checker for device code, not a CPU path of the product."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import oracle
from util import contact_lcp, have_ref

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pd, pi, pu8 = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
OL = oracle._lib()
OL.nbo_lcp_cascade.argtypes = [C.c_int, pd, pd, pd, pd, pd, pi, C.c_double, pd, C.POINTER(C.c_uint32), pd]
CFM = 1e-4
STAGE_BITS = 0x4 | 0x8 | 0x10 | 0x20 | 0x40


def _p(a):
    return a.ctypes.data_as(pd)


def _pi(a):
    return a.ctypes.data_as(pi)


def _pu(a):
    return a.ctypes.data_as(pu8) if a is not None else None


def _build_gen_shim(maxc):
    src = os.path.join(HERE, "host_shim", "gen_shim.cpp")
    out = os.path.join(HERE, "host_shim", "libgen_shim.so" if maxc == 64 else f"libgen_shim{maxc}.so")
    deps = [src] + [os.path.join(ROOT, "nimblephysics_amd", "csrc", f) for f in ("gen_lcp_dev.hpp", "gen_dantzig_dev.hpp", "lcp_dev.hpp", "spatial_dev.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-pthread", f"-DNBL_MAXC={maxc}", "-I", os.path.join(HERE, "host_shim"),
                               "-I", os.path.join(ROOT, "nimblephysics_amd", "csrc"), "-o", out, src])
    lib = C.CDLL(out)
    assert lib.gshim_rows() == 3 * maxc
    lib.gshim_cascade.argtypes = [C.c_int, pd, pd, pd, pu8, pu8, pu8, pd, C.c_double, pd, pd, pi]
    return lib


@pytest.fixture(scope="module")
def gen():
    return _build_gen_shim(64)


@pytest.fixture(scope="module")
def gen128():
    """the same device text compiled with the 384-row cap (the instantiation a model gets when it asks for 65 .. 128 contact slots)"""
    return _build_gen_shim(128)


def test_pinv_equals_numpy_pinv_up_to_192_rows(gen):
    """Full-rank and rank-deficient, symmetric and non-symmetric, rows / columns masked out (zero) the way the kernels select the clamping
    block: rank exact, Q^+ to 1e-9 (times the size)."""
    rng = np.random.default_rng(0)
    for trial, m in enumerate([6, 24, 24, 48, 60, 60, 96, 120, 120, 192, 192]):
        c = int(rng.integers(max(1, m // 2), m + 1)); k = int(rng.integers(1, c + 1)) if trial % 3 else c
        idx = np.sort(rng.choice(m, c, replace=False))
        U = rng.normal(0, 1, (c, k)); V = rng.normal(0, 1, (c, k))
        sub = U @ U.T if trial % 2 == 0 else U @ V.T
        Q = np.zeros((m, m)); Q[np.ix_(idx, idx)] = sub
        P = np.zeros((m, m))
        rank = gen.gshim_pinv(m, _p(np.ascontiguousarray(Q)), c, _p(P))
        ref = np.linalg.pinv(Q, rcond=1e-11)
        assert rank == k, (m, c, k, rank)
        assert np.abs(P - ref).max() <= 1e-9 * m * max(np.abs(ref).max(), 1e-30), (m, c, k, np.abs(P - ref).max() / np.abs(ref).max())
    Z = np.zeros((30, 30)); P = np.ones((30, 30))
    assert gen.gshim_pinv(30, _p(Z), 5, _p(P)) == 0 and not P.any()


def test_pinv_of_exactly_singular_symmetric_blocks_never_inverts_round_off(gen):
    """Small symmetric positive semi-definite Q that are EXACTLY rank deficient (two contacts at one point, four coplanar corners: the
    Delassus block J M^-1 J^T of dependent rows): the trailing pivot of the factorisation is pure round-off, a few eps of the largest.  With
    the reference's eps * size threshold such a pivot passed on the device in two worlds of 1.2 M (soak seeds 240049 / 243083 on the general
    build: a 3 x 3 Q of rank 2 inverted at rank 3, gradients of 1e13; tools/dbg/general_soak_case.py); with the symmetric policy (64 x) the rank
    must never exceed the true one and Q^+ must be numpy's.  5000 random cases of 3 - 12 rows (the count printed at the end is how often the
    1 x threshold would have failed on THESE cases: rare enough to be 0 here)."""
    rng = np.random.default_rng(11)
    inverted_round_off = 0
    for trial in range(5000):
        m = int(rng.integers(3, 13)); k = int(rng.integers(1, m))
        J = rng.normal(0, 1, (k, 8))
        mix = rng.integers(-2, 3, (m, k)).astype(float)
        mix[:k] = np.eye(k)                                       # rows k.. are integer combinations of the first k: exact dependence up to the rounding of J M J^T
        Jf = mix @ J
        Q = Jf @ np.diag(rng.uniform(0.1, 10, 8)) @ Jf.T
        Q = 0.5 * (Q + Q.T)
        P = np.zeros((m, m))
        rank = gen.gshim_pinv_sym(m, _p(np.ascontiguousarray(Q)), m, _p(P))
        assert rank <= k, (trial, m, k, rank)
        ref = np.linalg.pinv(Q, rcond=1e-10)
        assert np.abs(P - ref).max() <= 1e-7 * max(np.abs(ref).max(), 1e-30), (trial, m, k, rank)
        P2 = np.zeros((m, m))
        inverted_round_off += gen.gshim_pinv(m, _p(np.ascontiguousarray(Q)), m, _p(P2)) > k
    print(f"[singular symmetric Q] with the reference's eps * size threshold {inverted_round_off} of 5000 factorisations invert a round-off pivot")


def test_pinv_of_a_tower_of_flat_contacts(gen):
    """What a tower of cubes produces: blocks of four coplanar corners (a 12-row block of rank 6 per interface), coupled through the
    cubes' inertias; friction rows on their bound folded into their normal's column (non-symmetric)."""
    rng = np.random.default_rng(5)
    for ncubes in (3, 6, 10):
        nc = 4 * ncubes; m = 3 * nc; ndof = 6 * ncubes
        J = np.zeros((m, ndof))
        for k in range(ncubes):                              # interface k acts on cube k (and on cube k - 1 with the other sign)
            for c in range(4):
                p = np.array([(-1) ** c * 0.1, -0.1, (-1) ** (c // 2) * 0.1])
                for d in range(3):
                    e = np.eye(3)[[1, 0, 2][d]]
                    row = 3 * (4 * k + c) + d
                    J[row, 6 * k:6 * k + 6] = np.concatenate([np.cross(p, e), e])
                    if k > 0:
                        pb = p + np.array([0, 0.2, 0])
                        J[row, 6 * (k - 1):6 * (k - 1) + 6] = -np.concatenate([np.cross(pb, e), e])
        A = J @ np.diag(rng.uniform(0.5, 2, ndof)) @ J.T
        Q = A.copy()
        ub = rng.choice(np.arange(m)[np.arange(m) % 3 != 0], m // 6, replace=False)
        keep = np.ones(m, bool); keep[ub] = False
        for u in ub:
            Q[:, u - u % 3] += rng.choice([-1.0, 1.0]) * A[:, u]
        Q[~keep, :] = 0; Q[:, ~keep] = 0
        P = np.zeros((m, m))
        rank = gen.gshim_pinv(m, _p(np.ascontiguousarray(Q)), int(keep.sum()), _p(P))
        ref = np.linalg.pinv(Q, rcond=1e-10)
        assert rank == np.linalg.matrix_rank(Q, tol=1e-9 * np.abs(Q).max())
        err = np.abs(P - ref).max() / np.abs(ref).max()
        print(f"[tower pinv] {ncubes} cubes, {m} rows, rank {rank}: {err:.1e}")
        assert err < 1e-8


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_dantzig_is_bit_identical_to_the_reference_dsolvelcp_up_to_192_rows(gen):
    rng = np.random.default_rng(0)
    solved = failed = 0
    sizes = [1, 2, 3, 5, 8, 8, 11, 16, 16, 20, 27, 32, 40, 48, 64, 64]
    for trial in range(60):
        nc = sizes[trial % len(sizes)]; n = 3 * nc
        ndof = n + int(rng.integers(0, 6)) if trial % 3 == 0 else int(rng.choice([3, 6, 12, 30, 60]))   # 2 of 3: rank-deficient A
        A, b, lo, hi, fi = contact_lcp(rng, nc, ndof)
        if trial % 4 == 1:
            b = np.abs(b)                                      # more rows end in the clamping set: long factor, many removals
        xr = np.zeros(n); xd = np.zeros(n)
        okr = OL.nbo_lcp_dantzig(n, _p(A), _p(xr), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), _pi(fi.copy()), 1)
        okd = gen.gshim_dantzig(n, _p(A), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), _pi(fi.copy()), _p(xd))
        if okd == -1:
            assert okr == 0 or not np.all(np.isfinite(xr)), (trial, okr)
            continue
        assert okr == okd, (trial, n, ndof, okr, okd)
        if okr == 1:
            solved += 1
            assert np.array_equal(xr, xd), (trial, n, ndof, np.abs(xr - xd).max())
        else:
            failed += 1
    print(f"Dantzig up to 192 rows: {solved} solved bit for bit, {failed} early exits, all flags equal")
    assert solved > 20 and failed > 0


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("lanes", [1, 5, 64])
def test_the_wave_shared_dantzig_driver_is_bit_identical_to_the_reference_with_one_lane_and_with_many(gen, lanes):
    """genDantzigPar (round 6: what the device runs - the loops with independent iterations strided over the lanes, dDot's products in
    parallel and its running sum in index order, the triangular solves column block by column block) against the reference's compiled
    dSolveLCP: x and the success flag BIT FOR BIT, with ONE lane (plain sequential code), with 5 lanes (a count that divides nothing:
    every strided loop has a ragged end) and with 64 (the wavefront) - threads and barriers, so a missing barrier or a racing in-place
    shift shows as a mismatch."""
    rng = np.random.default_rng(10 + lanes)
    solved = failed = 0
    sizes = [1, 2, 3, 5, 8, 8, 11, 16, 16, 20, 27, 32, 40, 48, 64, 64] if lanes == 1 else [1, 2, 3, 5, 8, 8, 11, 16, 20, 27, 40, 64]
    for trial in range(60 if lanes == 1 else (36 if lanes < 64 else 18)):      # (64 threads on a few cores: barriers are slow)
        nc = sizes[trial % len(sizes)]; n = 3 * nc
        ndof = n + int(rng.integers(0, 6)) if trial % 3 == 0 else int(rng.choice([3, 6, 12, 30, 60]))   # 2 of 3: rank-deficient A
        A, b, lo, hi, fi = contact_lcp(rng, nc, ndof)
        if trial % 4 == 1:
            b = np.abs(b)                                      # more rows end in the clamping set: long factor, many removals
        xr = np.zeros(n); xd = np.zeros(n); xs = np.zeros(n)
        okr = OL.nbo_lcp_dantzig(n, _p(A), _p(xr), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), _pi(fi.copy()), 1)
        okd = gen.gshim_dantzig_par(n, _p(A), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), _pi(fi.copy()), _p(xd), lanes)
        oks = gen.gshim_dantzig(n, _p(A), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), _pi(fi.copy()), _p(xs))
        assert okd == oks and np.array_equal(xd, xs), (trial, n, lanes, okd, oks, "the wave-shared driver differs from the sequential one")
        if okd == -1:
            assert okr == 0 or not np.all(np.isfinite(xr)), (trial, okr)
            continue
        assert okr == okd, (trial, n, ndof, okr, okd)
        if okr == 1:
            solved += 1
            assert np.array_equal(xr, xd), (trial, n, ndof, np.abs(xr - xd).max())
        else:
            failed += 1
    print(f"wave-shared Dantzig, {lanes} lane(s), up to 192 rows: {solved} solved bit for bit, {failed} early exits, all flags equal")
    assert solved > 10 and failed > 0


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_the_384_row_build_of_the_same_text_dantzig_bit_identical_and_pinv_up_to_384_rows(gen128):
    """-DNBL_MAXC=128: nothing but the cap of the per-row arrays changes (the leading dimension of the matrices is a run-time property of the
    problem).  Dantzig bit-identical to the reference's dSolveLCP at 65 .. 128 contacts, the pseudo-inverse against numpy up to 384 rows."""
    rng = np.random.default_rng(3)
    solved = 0
    for trial, nc in enumerate([65, 72, 80, 96, 110, 128, 128]):
        n = 3 * nc
        ndof = n + 3 if trial % 3 == 0 else int(rng.choice([30, 60, 120]))
        A, b, lo, hi, fi = contact_lcp(rng, nc, ndof)
        if trial % 2 == 1:
            b = np.abs(b)
        xr = np.zeros(n); xd = np.zeros(n)
        okr = OL.nbo_lcp_dantzig(n, _p(A), _p(xr), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), _pi(fi.copy()), 1)
        okd = gen128.gshim_dantzig(n, _p(A), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), _pi(fi.copy()), _p(xd))
        if okd == -1:
            assert okr == 0 or not np.all(np.isfinite(xr)), (trial, okr)
            continue
        assert okr == okd, (trial, n, okr, okd)
        if okr == 1:
            solved += 1
            assert np.array_equal(xr, xd), (trial, n, np.abs(xr - xd).max())
    assert solved >= 2
    for trial, m in enumerate([200, 264, 384, 384]):
        c = int(rng.integers(m // 2, m + 1)); k = c if trial % 2 == 0 else int(rng.integers(c // 2, c))
        idx = np.sort(rng.choice(m, c, replace=False))
        U = rng.normal(0, 1, (c, k)); V = rng.normal(0, 1, (c, k))
        Q = np.zeros((m, m)); Q[np.ix_(idx, idx)] = U @ U.T if trial < 2 else U @ V.T
        P = np.zeros((m, m))
        rank = gen128.gshim_pinv(m, _p(np.ascontiguousarray(Q)), c, _p(P))
        ref = np.linalg.pinv(Q, rcond=1e-11)
        assert rank == k, (m, c, k, rank)
        assert np.abs(P - ref).max() <= 1e-9 * m * max(np.abs(ref).max(), 1e-30), (m, c, k)


def _contact_problem(rng, trial, R):
    from test_coop_host import _contact_problem as cp
    return cp(rng, trial, R)


@pytest.mark.parametrize("maxc", [8, 16])
def test_stage0_equals_the_wavefront_cooperative_code(gen, maxc):
    """resting, sliding (upper-bound rows), perturbed and random contact problems with rank-deficient A, cold and warm started: the same
    accept / reject decision, pre-solve x, x, row classes as the device code of the 24- / 48-row builds (host wave emulation)."""
    from test_coop_host import _build_shim
    shim = _build_shim(maxc)
    R = shim.R
    rng = np.random.default_rng(1)
    n_ok = n_fail = n_ub = 0
    for trial in range(150 if R == 24 else 45):
        m, A, b, mu, have, xc = _contact_problem(rng, trial, R)
        X1 = np.zeros(R); X01 = np.zeros(R); c1 = np.zeros(R, np.int32); E1 = np.zeros(R); P = np.zeros((R, R))
        r1 = shim.shim_coop_stage0(m, _p(A), _p(b), _p(mu), have, _p(xc), _p(X1), _p(X01), _pi(c1), _p(E1), _p(P))
        Am = np.ascontiguousarray(A[:m, :m])
        X2 = np.zeros(m); X02 = np.zeros(m); c2 = np.zeros(m, np.int32); E2 = np.zeros(m); P2 = np.zeros((m, m))
        r2 = gen.gshim_stage0(m, _p(Am), _p(b[:m].copy()), _p(mu), None, None, None, have, _p(xc[:m].copy()), _p(X2), _p(X02), _pi(c2), _p(E2), _p(P2))
        assert (r1 & 1) == (r2 & 1), (trial, r1, r2)
        assert np.abs(X01[:m] - X02).max() <= 1e-9 * max(1.0, np.abs(X02).max())
        if not (r1 & 1):
            n_fail += 1
            continue
        n_ok += 1
        assert np.abs(X1[:m] - X2).max() <= 1e-9 * max(1.0, np.abs(X2).max())
        assert np.array_equal(c1[:m], c2) and np.array_equal(E1[:m], E2)
        n_ub += int((c2 == 2).any())
        if (r1 & 2) and (r2 & 2):
            assert np.abs(P[:m, :m] - P2).max() <= 1e-7 * max(np.abs(P2).max(), 1e-30)
    assert (n_ok > 10 and n_fail > 10 and n_ub > 2) if R == 24 else (n_ok > 3 and n_fail > 10), (n_ok, n_fail, n_ub)


def _oracle_cascade(n, A, x0, b, lo, hi, fi, cfm=CFM):
    x = np.zeros(n); st = C.c_uint32(0); cfm_used = C.c_double(0)
    valid = OL.nbo_lcp_cascade(n, _p(np.ascontiguousarray(A)), _p(np.ascontiguousarray(x0)), _p(np.ascontiguousarray(b)), _p(np.ascontiguousarray(lo)),
                               _p(np.ascontiguousarray(hi)), _pi(fi), cfm, _p(x), C.byref(st), C.byref(cfm_used))
    return x, st.value, cfm_used.value, bool(valid)


def _gen_cascade(gen, n, A, x0, b, hi, mask=None):
    mu = np.ascontiguousarray(hi[1::3])
    X = np.zeros(n); cls = np.zeros(n, np.int32); cfm = C.c_double(0)
    st = gen.gshim_cascade(n, _p(np.ascontiguousarray(A)), _p(np.ascontiguousarray(b)), _p(mu), _pu(mask), None, None, _p(np.ascontiguousarray(x0)), CFM, _p(X), C.byref(cfm), _pi(cls))
    return X, st, cfm.value, cls


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_cascade_equals_the_oracles_on_the_references_fixtures_and_on_big_problems(gen):
    """stages 1-3 in the reference's order of preference: the same stage and the same x as the oracle's cascade (whose stage 1 IS the
    reference's dSolveLCP) on the reference's own fixtures and on random contact problems of 8 .. 64 contacts, rank-deficient ones
    included.  (The raw cascade solution is compared: stage bits and x before the standardisation, which the kernels' record and
    the oracle's CGGM restate separately.)"""
    FIX = json.load(open(os.path.join(HERE, "golden", "lcp_fixtures.json")))
    rng = np.random.default_rng(3)
    cases = []
    for name in sorted(FIX):
        f = FIX[name]
        n = len(f["b"])
        if n % 3:
            continue
        fi = np.ascontiguousarray(f["fIndex"], dtype=np.int32)
        if not (np.all(fi[0::3] == -1) and np.all(fi[1::3] == np.arange(0, n, 3)) and np.all(fi[2::3] == np.arange(0, n, 3))):
            continue
        cases.append((name, n, np.array(f["A"], float).reshape(n, n), np.array(f.get("x", np.zeros(n)), float), np.array(f["b"], float),
                      np.array(f["lo"], float), np.array(f["hi"], float), fi))
    for trial in range(30):
        nc = [8, 12, 16, 20, 27, 40, 64][trial % 7]
        ndof = int(rng.choice([12, 30, 60])) if trial % 2 else 3 * nc + 3
        A, b, lo, hi, fi = contact_lcp(rng, nc, ndof)
        cases.append((f"random{trial}", 3 * nc, A, rng.normal(0, 0.05, 3 * nc) * (trial % 3 == 0), b, lo, hi, fi))
    stages = {}
    for name, n, A, x0, b, lo, hi, fi in cases:
        xo, sto, cfmo, valid = _oracle_cascade(n, A, x0, b, lo, hi, fi)
        xg, stg, cfmg, cls = _gen_cascade(gen, n, A, x0, b, hi)
        assert (stg & STAGE_BITS) == (sto & STAGE_BITS), (name, hex(stg), hex(sto))
        assert cfmg == cfmo
        stages[sto & STAGE_BITS] = stages.get(sto & STAGE_BITS, 0) + 1
        if stg & 0x100:       # standardised: the impulses are the least-squares solution on the final classification, not the raw iterate
            continue
        assert np.abs(xg - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max()), (name, np.abs(xg - xo).max())
    print("cascade stages taken:", {hex(k): v for k, v in stages.items()})
    assert len(stages) >= 2


def test_cascade_on_one_constrained_group_equals_the_cascade_of_that_group_alone(gen):
    rng = np.random.default_rng(11)
    for trial in range(12):
        ncA, ncB = int(rng.integers(2, 20)), int(rng.integers(2, 20))
        parts = [contact_lcp(rng, nc, int(rng.choice([6, 12, 3 * nc + 2]))) for nc in (ncA, ncB)]
        mA, mB = 3 * ncA, 3 * ncB
        m = mA + mB
        A = np.zeros((m, m)); A[:mA, :mA] = parts[0][0]; A[mA:, mA:] = parts[1][0]
        b = np.concatenate([parts[0][1], parts[1][1]]); hi = np.concatenate([parts[0][3], parts[1][3]])
        off = 0
        for g, (Ag, bg, log, hig, fig) in enumerate(parts):
            mg = len(bg)
            mask = np.zeros(m, np.uint8); mask[off:off + mg] = 1
            X, st, cfm, cls = _gen_cascade(gen, m, A, np.zeros(m), b, hi, mask)
            X1, st1, cfm1, cls1 = _gen_cascade(gen, mg, Ag, np.zeros(mg), bg, hig)
            assert st == st1 and cfm == cfm1, (trial, g, hex(st), hex(st1))
            assert np.array_equal(cls[off:off + mg], cls1)
            assert np.abs(X[off:off + mg] - X1).max() <= 1e-10 * max(1.0, np.abs(X1).max())
            assert not X[:off].any() and not X[off + mg:].any()
            off += mg


@pytest.mark.parametrize("lanes", [7, 64])
def test_the_stages_with_the_work_shared_by_many_lanes_equal_the_one_lane_run_bit_for_bit(gen, lanes):
    """Stage 1 (load, the shared column-pair search of reduce, the wave-shared Dantzig driver), stage 2 (CFM, reduce, Gauss-Seidel in
    residual form: with 64 lanes every lane HOLDS its rows in registers and fetches a step's column one step ahead, with 7 lanes and with
    one everything is read in place) and stage 3 (the friction rows dropped in one gather) of gen_lcp_dev.hpp / gen_dantzig_dev.hpp on
    threads + barriers against the same text under the one-lane policy: flags and candidate x BIT FOR BIT - every number is formed by one
    lane with the same operands in the same order whatever the lane count.  Random contact problems of 2 .. 40 contacts, rank-deficient
    ones, merged columns (two contacts at one point) included."""
    rng = np.random.default_rng(100 + lanes)
    gen.gshim_stage.argtypes = [C.c_int, C.c_int, pd, pd, pd, pu8, pd, C.c_double, pd]
    gen.gshim_stage_lanes.argtypes = [C.c_int, C.c_int, pd, pd, pd, pu8, pd, C.c_double, pd, C.c_int]
    seen = {1: set(), 2: set(), 3: set()}
    for trial in range(18 if lanes < 64 else 9):
        nc = [2, 5, 8, 8, 12, 20, 27, 40, 16][trial % 9]
        ndof = int(rng.choice([6, 12, 30])) if trial % 2 else 3 * nc + 3
        A, b, lo, hi, fi = contact_lcp(rng, nc, ndof)
        m = 3 * nc
        if trial % 5 == 4 and nc >= 2:                     # two contacts at the same point: identical rows / columns, reduce merges them
            A[3:6, :] = A[0:3, :]; A[:, 3:6] = A[:, 0:3]; b[3:6] = b[0:3]; hi[3:6] = hi[0:3]
        mu = np.ascontiguousarray(hi[1::3])
        x0 = rng.normal(0, 0.05, m) * (trial % 3 == 0)
        for stage in (1, 2, 3):
            X1 = np.zeros(m); XL = np.zeros(m)
            f1 = gen.gshim_stage(stage, m, _p(np.ascontiguousarray(A)), _p(b.copy()), _p(mu), None, _p(x0.copy()), 1e-3, _p(X1))
            fl = gen.gshim_stage_lanes(stage, m, _p(np.ascontiguousarray(A)), _p(b.copy()), _p(mu), None, _p(x0.copy()), 1e-3, _p(XL), lanes)
            assert fl == f1, (trial, stage, lanes, fl, f1)
            assert np.array_equal(X1, XL), (trial, stage, lanes, np.abs(X1 - XL).max())
            seen[stage].add(int(f1))
    print(f"stages on {lanes} lanes == one lane, flags seen:", seen)
    assert len(seen[1]) >= 2 and len(seen[2]) >= 1


@pytest.mark.parametrize("model_rows", [24, 48, 88])
def test_the_device_placement_of_the_scratch_changes_no_bit(gen, model_rows):
    """k_contact_solve_gen keeps the cascade's sixteen vectors in LDS (a pool of max(16 rows, 1152) doubles for a model of `rows` rows) and
    lends that pool, packed by the WORLD's rows, to the working pair of the pseudo-inverse (genPinvPair) and to the matrix of the
    Gauss-Seidel sweeps (genPgsAT, in the place of the Dantzig driver's vectors).  Placement only: the whole cascade (stage 0, the stages,
    standardisation) and every stage on 64 lanes with the device's layout (gshim_device_pool) equal the plain layout bit for bit - worlds
    of 2 .. 29 contacts in models of 8 .. 29 slots, the pool too small for the pair / the matrix included."""
    rng = np.random.default_rng(500 + model_rows)
    gen.gshim_device_pool.argtypes = [C.c_int]; gen.gshim_device_pool.restype = None
    gen.gshim_stage_lanes.argtypes = [C.c_int, C.c_int, pd, pd, pd, pu8, pd, C.c_double, pd, C.c_int]
    try:
        for trial in range(4):
            nc = int(rng.integers(2, model_rows // 3 + 1))
            ndof = int(rng.choice([6, 12, 30])) if trial % 2 else 3 * nc + 3
            A, b, lo, hi, fi = contact_lcp(rng, nc, ndof)
            m = 3 * nc
            mu = np.ascontiguousarray(hi[1::3])
            x0 = rng.normal(0, 0.05, m) * (trial % 3 == 0)
            res = []
            for pool in (0, model_rows):
                gen.gshim_device_pool(pool)
                X, st, cfm, cls = _gen_cascade(gen, m, A, x0, b, hi)
                stages = []
                for stage in (1, 2, 3):
                    XL = np.zeros(m)
                    fl = gen.gshim_stage_lanes(stage, m, _p(np.ascontiguousarray(A)), _p(b.copy()), _p(mu), None, _p(x0.copy()), 1e-3, _p(XL), 64)
                    stages.append((fl, XL))
                res.append((X, st, cfm, cls, stages))
            (X0, st0, cfm0, cls0, sg0), (X1, st1, cfm1, cls1, sg1) = res
            assert st0 == st1 and cfm0 == cfm1 and np.array_equal(cls0, cls1) and np.array_equal(X0, X1), (trial, nc, hex(st0), hex(st1))
            for (f0, x0_), (f1, x1_) in zip(sg0, sg1):
                assert f0 == f1 and np.array_equal(x0_, x1_), (trial, nc)
    finally:
        gen.gshim_device_pool(0)
