"""The oracle's restatement of LCPUtils (oracle/lcp.hpp: isLCPSolutionValid, reduceLcp, removeFrictionLcp) against the REFERENCE'S OWN
functions, compiled from dart/constraint/LCPUtils.cpp:12-80, 144-247, 346-549 where they lie into oracle/_ref/liblcputils_ref.so
(oracle/ref_build.py::build_lcputils; a small dynamic matrix class stands in for Eigen, which is not on this machine): the same verdicts,
and every number of the reduced problems and of mapOut BIT FOR BIT - on the eight literal fixtures of the reference's own
unittests/unit/test_LCPUtils.cpp and on random contact LCPs with duplicated contacts, near-duplicates on both sides of the merge
threshold, solutions on both sides of the validity tolerance, friction dropped or not.  Test infrastructure."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from util import contact_lcp

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(os.path.dirname(oracle.__file__), "_ref", "liblcputils_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/liblcputils_ref.so not built (python oracle/ref_build.py, needs /root/reference)")
FIX = json.load(open(os.path.join(HERE, "golden", "lcp_fixtures.json")))
pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(pd)


def _pi(a):
    return a.ctypes.data_as(pi)


def _libs():
    return oracle._lib(), C.CDLL(REF)


def _valid_both(A, x, b, lo, hi, fi, ignore):
    L, R = _libs()
    n = len(b)
    A, x, b, lo, hi = (_d(v) for v in (A, x, b, lo, hi))
    fi = np.ascontiguousarray(fi, dtype=np.int32)
    o = L.nbo_lcp_valid(n, _p(A), _p(x), _p(b), _p(lo), _p(hi), _pi(fi), int(ignore))
    r = R.ref_lcp_valid(n, _p(A), _p(x), _p(b), _p(hi), _p(lo), _pi(fi), int(ignore))
    return bool(o), bool(r)


def _reduce_both(A, x, b, lo, hi, fi, remove_friction):
    """-> (oracle result, reference result), each (nr, A, x, b, lo, hi, findex, mapOut)"""
    L, R = _libs()
    n = len(b)
    A, x, b, lo, hi = (_d(v) for v in (A, x, b, lo, hi))
    fi = np.ascontiguousarray(fi, dtype=np.int32)
    Ar, xr, br, lor, hir, fr, mo = np.zeros(n * n), np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n, np.int32), np.zeros(n * n)
    nr = L.nbo_lcp_reduce(n, _p(A), _p(x), _p(b), _p(lo), _p(hi), _pi(fi), int(remove_friction), _p(Ar), _p(xr), _p(br), _p(lor), _p(hir), _pi(fr), _p(mo))
    ora = (nr, Ar[:nr * nr].copy(), xr[:nr].copy(), br[:nr].copy(), lor[:nr].copy(), hir[:nr].copy(), fr[:nr].copy(), mo[:n * nr].copy())
    A2, x2, b2, lo2, hi2, f2, m2 = A.copy().reshape(-1), x.copy(), b.copy(), lo.copy(), hi.copy(), fi.copy(), np.zeros(n * n)
    fn = R.ref_lcp_remove_friction if remove_friction else R.ref_lcp_reduce
    nr2 = fn(n, _p(A2), _p(x2), _p(b2), _p(hi2), _p(lo2), _pi(f2), _p(m2))
    ref = (nr2, A2[:nr2 * nr2].copy(), x2[:nr2].copy(), b2[:nr2].copy(), lo2[:nr2].copy(), hi2[:nr2].copy(), f2[:nr2].copy(), m2[:n * nr2].copy())
    return ora, ref


def _assert_same(ora, ref, tag):
    assert ora[0] == ref[0], (tag, "reduced sizes", ora[0], ref[0])
    for name, a, b in zip(("A", "x", "b", "lo", "hi", "findex", "mapOut"), ora[1:], ref[1:]):
        assert np.array_equal(a, b), (tag, name, "not bit-identical")


def _fixture(name):
    f = FIX[name]
    n = len(f["b"])
    return _d(f["A"]).reshape(n, n), _d(f.get("x", np.zeros(n))), _d(f["b"]), _d(f["lo"]), _d(f["hi"]), np.ascontiguousarray(f["fIndex"], dtype=np.int32)


@pytest.mark.parametrize("name", sorted(FIX))
def test_the_references_own_fixtures(name):
    A, x, b, lo, hi, fi = _fixture(name)
    for rf in (False, True):
        ora, ref = _reduce_both(A, x, b, lo, hi, fi, rf)
        _assert_same(ora, ref, (name, "removeFriction" if rf else "reduce"))
    merged = _reduce_both(A, x, b, lo, hi, fi, False)[1][0]
    print(f"[{name}] {len(b)} rows -> reduce: {merged}, removeFriction: {_reduce_both(A, x, b, lo, hi, fi, True)[1][0]}")
    for ignore in (False, True):
        for xx in (x, np.zeros(len(b)), np.linalg.lstsq(A, b, rcond=None)[0]):
            o, r = _valid_both(A, xx, b, lo, hi, fi, ignore)
            assert o == r, (name, ignore)


def _duplicated_problem(rng, nc, ndof):
    """contact LCP in which some contacts repeat another one exactly (two corners of one face at one point: columns, b, bounds and findex
    pattern all equal - what reduce merges), some up to a perturbation on either side of the 1e-4 squared-distance threshold"""
    A, b, lo, hi, fi = contact_lcp(rng, nc, ndof)
    n = 3 * nc
    J = np.linalg.cholesky(A + 1e-9 * np.eye(n))      # any factor with A = J J^T
    kind = rng.integers(0, 4)
    src, dst = rng.choice(nc, 2, replace=False)
    if kind > 0:
        pert = {1: 0.0, 2: 3e-3, 3: 3e-2}[int(kind)]                 # exact copy / below the threshold / above it
        J[3 * dst:3 * dst + 3] = J[3 * src:3 * src + 3] + pert * rng.normal(0, 1, (3, n)) / np.sqrt(n)
        b[3 * dst:3 * dst + 3] = b[3 * src:3 * src + 3] + (pert * 1e-2 if kind < 3 else 2e-4)
        hi[3 * dst:3 * dst + 3] = hi[3 * src:3 * src + 3]; lo[3 * dst:3 * dst + 3] = lo[3 * src:3 * src + 3]
    A = J @ J.T
    return _d(A), b, lo, hi, fi


def test_reduce_and_remove_friction_bit_for_bit_on_random_contact_lcps():
    rng = np.random.default_rng(17)
    sizes = {}
    n_merged = 0
    for trial in range(400):
        nc = int(rng.integers(2, 9))
        A, b, lo, hi, fi = _duplicated_problem(rng, nc, int(rng.integers(3, 3 * nc + 3)))
        x = rng.normal(0, 1, 3 * nc)
        for rf in (False, True):
            ora, ref = _reduce_both(A, x, b, lo, hi, fi, rf)
            _assert_same(ora, ref, (trial, rf))
            if not rf and ref[0] < 3 * nc:
                n_merged += 1
            sizes[(3 * nc, rf)] = ref[0]
    assert n_merged > 60, n_merged        # the merge path is really exercised (normal rows of duplicated contacts; friction rows follow their findex)
    print(f"400 random contact LCPs (6 .. 24 rows): {n_merged} with merged columns; reduce / removeFriction bit-identical to the reference's")


def test_validity_verdicts_on_both_sides_of_the_tolerance():
    rng = np.random.default_rng(23)
    agree = {True: 0, False: 0}
    for trial in range(600):
        nc = int(rng.integers(1, 9))
        A, b, lo, hi, fi = contact_lcp(rng, nc, int(rng.integers(3, 3 * nc + 3)))
        n = 3 * nc
        # a solution from the oracle's solver cascade (valid in most cases), then nudged by 0 / 0.3 / 3 validity tolerances
        x = np.zeros(n)
        st = C.c_uint32(0); cfm = C.c_double(0)
        oracle._lib().nbo_lcp_cascade(n, _p(A), _p(np.zeros(n)), _p(b), _p(lo), _p(hi), _pi(fi), C.c_double(1e-4), _p(x), C.byref(st), C.byref(cfm))
        x += rng.choice([0.0, 3e-6, 3e-5]) * rng.normal(0, 1, n)
        for ignore in (False, True):
            xx = x.copy()
            if ignore and rng.random() < 0.7:
                xx[fi >= 0] = 0.0
            o, r = _valid_both(A, xx, b, lo, hi, fi, ignore)
            assert o == r, (trial, ignore)
            agree[o] += 1
    assert agree[True] > 50 and agree[False] > 50, agree
    print(f"isLCPSolutionValid: {agree[True]} valid / {agree[False]} invalid verdicts, all equal to the reference's")
