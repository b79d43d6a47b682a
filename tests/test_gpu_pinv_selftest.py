"""The device pseudo-inverses run ON THE GPU through the C ABI (nbl_selftest_pinv): coopPinv (Householder QR + complete orthogonal
decomposition, the role of Eigen's completeOrthogonalDecomposition in CGGM.cpp:280 / LCPUtils.cpp:113) and coopPinvSym (two
Cholesky factorisations with the GEMMs on the matrix cores, the route of every symmetric positive semi-definite Q) against numpy:
rank exact, Q^+ to cond(Q) eps, the two routes against each other.  (The same code on the host wave emulation: tests/test_coop_host.py.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases(rng, count, R=24):
    Q = np.zeros((count, R, R)); size = np.zeros(count, np.int32); rank = np.zeros(count, np.int32)
    for t in range(count):
        c = int(rng.integers(1, R + 1)); idx = np.sort(rng.choice(R, c, replace=False))
        kind = t % 6
        if kind == 0:                                   # a standing robot: c rows on two 6-DOF bodies
            k = min(c, 12); J = rng.normal(0, 1, (c, 12)); sub = J @ np.diag(rng.uniform(0.1, 2, 12)) @ J.T
        elif kind == 1:                                 # ... with the fallback CFM: full rank, ill conditioned
            k = c; J = rng.normal(0, 1, (c, 12)); sub = J @ np.diag(rng.uniform(0.1, 2, 12)) @ J.T + 1e-4 * np.eye(c)
        elif kind == 2:                                 # full rank, well conditioned (two contacts of a small body: the leading block only)
            c = min(c, 6); idx = np.arange(c); k = c; U = rng.normal(0, 1, (c, c + 3)); sub = U @ U.T
        else:
            k = int(rng.integers(1, c + 1)); U = rng.normal(0, 1, (c, k)); sub = U @ U.T
        Q[t][np.ix_(idx, idx)] = 0.5 * (sub + sub.T); size[t] = c; rank[t] = k
    return Q, size, rank


@pytest.mark.parametrize("R", [24, 48])      # the 24-row and the 48-row instantiation of the library (csrc/abi_variants.h)
def test_device_pseudo_inverses_equal_numpy_pinv(R):
    from nimblephysics_amd._lib import check, lib
    rng = np.random.default_rng(11 + R)
    count = 768 if R == 24 else 384
    Q, size, rank = _cases(rng, count, R)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    out = {}
    for route in (0, 1):
        P = np.zeros_like(Q); r = np.zeros(count, np.int32)
        check(lib().nbl_selftest_pinv_rows(count, R, vp(Q), vp(size), route, vp(P), vp(r), 1, None), "nbl_selftest_pinv_rows")
        out[route] = (P, r)
    worst = {0: 0.0, 1: 0.0}
    for t in range(count):
        sv = np.linalg.svd(Q[t], compute_uv=False)
        k = rank[t]; cond = sv[0] / sv[k - 1]
        ref = np.linalg.pinv(Q[t], rcond=0.5 * sv[k - 1] / sv[0])
        scale = np.abs(ref).max()
        for route in (0, 1):
            P, r = out[route]
            assert r[t] == k, (route, t, r[t], k)
            e = np.abs(P[t] - ref).max() / scale
            worst[route] = max(worst[route], e / (cond * 2.2e-16))
            assert e <= 500 * cond * 2.2e-16, (route, t, e, cond)
    print("worst error in units of cond(Q) eps: Householder route", worst[0], " Cholesky route", worst[1])


@pytest.mark.parametrize("R", [24, 48])
def test_round_off_pivots_of_exactly_singular_contact_matrices_are_rejected_like_the_references_cod(R):
    """An exactly singular Q (two contacts of one body that span five directions): the last pivot of the diagonally pivoted Cholesky
    is pure round-off but comes out ABOVE the eps * size threshold of the reference's decomposition (1.4e-15 of the first pivot against
    1.3e-15), the COD's last |R_kk| below it (3e-16): tests/golden/pinv_rank_borderline.json, a matrix the randomised soak found
    (seed 30168).  Both device routes must say rank 5 and return the rank-5 pseudo-inverse; plus 256 synthetic matrices with exactly
    repeated / linearly dependent rows."""
    import json
    import os
    from nimblephysics_amd._lib import check, lib
    f = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pinv_rank_borderline.json")))
    A = np.array([[float.fromhex(x) for x in row] for row in f["A"]])
    rng = np.random.default_rng(12)
    mats, ranks = [A], [5]
    for t in range(256):
        nc = int(rng.integers(2, R // 3 + 1)); ndof = int(rng.choice([6, 7, 12]))
        J = rng.normal(0, 1, (3 * nc, ndof))
        if t % 2 == 0:
            J[3:6] = J[0:3] + (0.0 if t % 4 == 0 else 1.0) * rng.normal(0, 1, (1, ndof))     # a repeated contact / a contact one direction away
        M = J @ np.diag(rng.uniform(0.1, 2, ndof)) @ J.T
        mats.append(0.5 * (M + M.T)); ranks.append(int(np.linalg.matrix_rank(J)))
    count = len(mats)
    Q = np.zeros((count, R, R)); size = np.zeros(count, np.int32)
    for t, M in enumerate(mats):
        Q[t, :len(M), :len(M)] = M; size[t] = len(M)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    for route in (0, 1):
        P = np.zeros_like(Q); r = np.zeros(count, np.int32)
        check(lib().nbl_selftest_pinv_rows(count, R, vp(Q), vp(size), route, vp(P), vp(r), 1, None), "nbl_selftest_pinv_rows")
        assert np.array_equal(r, np.array(ranks)), (route, np.where(r != np.array(ranks))[0][:10])
        for t in (0, 1, 2, 3):
            sv = np.linalg.svd(Q[t], compute_uv=False)
            ref = np.linalg.pinv(Q[t], rcond=0.5 * sv[ranks[t] - 1] / sv[0])
            assert np.abs(P[t] - ref).max() <= 1e-9 * np.abs(ref).max(), (route, t)
