"""The PRODUCT's per-lane LCP device code (nimblephysics_amd/csrc/{lcp,dantzig}_dev.hpp) compiled for the host
through a stub hip_runtime.h (tests/host_shim) and checked against
  * the reference's own Dantzig solver dSolveLCP (oracle/_ref, compiled from dart/external/odelcpsolver),
  * the oracle's restated PGS / reduce / removeFriction (LCPUtils.cpp, PgsBoxedLcpSolver.cpp),
  * numpy's pseudo-inverse for the complete-orthogonal-decomposition solves (Eigen stand-in).
This is a checker for device code, not a CPU path of the product."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
L = oracle._lib()


@pytest.fixture(scope="module")
def shim():
    src = os.path.join(HERE, "host_shim", "lcp_shim.cpp")
    out = os.path.join(HERE, "host_shim", "liblcp_shim.so")
    deps = [src] + [os.path.join(ROOT, "nimblephysics_amd", "csrc", f) for f in ("lcp_dev.hpp", "dantzig_dev.hpp", "spatial_dev.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(HERE, "host_shim"),
                               "-I", os.path.join(ROOT, "nimblephysics_amd", "csrc"), "-o", out, src])
    return C.CDLL(out)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(pd)


def contact_lcp(rng, nc, ndof, cfm=0.0):
    n = 3 * nc
    J = rng.normal(0, 1, (n, ndof))
    A = J @ np.diag(rng.uniform(0.1, 2, ndof)) @ J.T + cfm * np.eye(n)
    b = rng.normal(0, 1, n) * rng.choice([1, 0.01])
    mu = rng.choice([0.5, 1.0], nc)
    lo = np.zeros(n); hi = np.full(n, np.inf); fi = np.full(n, -1, np.int32)
    for c in range(nc):
        for k in (1, 2):
            lo[3 * c + k] = -mu[c]; hi[3 * c + k] = mu[c]; fi[3 * c + k] = 3 * c
    return _d(A), _d(b), lo, hi, fi


def have_ref():
    return os.path.exists(os.path.join(os.path.dirname(oracle.__file__), "_ref", "libodelcp_ref.so"))


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_device_dantzig_equals_reference_dsolvelcp_full_rank(shim):
    rng = np.random.default_rng(0)
    for _ in range(400):
        nc = int(rng.integers(1, 9)); n = 3 * nc
        A, b, lo, hi, fi = contact_lcp(rng, nc, n + int(rng.integers(0, 6)))
        xr = np.zeros(n); xd = np.zeros(n)
        okr = L.nbo_lcp_dantzig(n, _p(A), _p(xr), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), fi.copy().ctypes.data_as(pi), 1)
        okd = shim.shim_dantzig(n, _p(A), _p(xd), _p(b), _p(lo), _p(hi), fi.ctypes.data_as(pi))
        assert okr == okd
        if okr == 1:
            assert np.allclose(xr, xd, rtol=1e-9, atol=1e-12)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_device_dantzig_vs_reference_on_rank_deficient_contact_problems(shim):
    """6-DOF body with up to 8 frictional contacts: A(C,C) goes singular; where the s <= 0 early-termination test is
    decided by round-off the two implementations may disagree (measured ~1.6 %); everywhere else x is identical."""
    rng = np.random.default_rng(1)
    agree = both = 0
    N = 600
    for _ in range(N):
        nc = int(rng.integers(2, 9)); n = 3 * nc
        A, b, lo, hi, fi = contact_lcp(rng, nc, 6)
        xr = np.zeros(n); xd = np.zeros(n)
        okr = L.nbo_lcp_dantzig(n, _p(A), _p(xr), _p(b.copy()), _p(lo.copy()), _p(hi.copy()), fi.copy().ctypes.data_as(pi), 1)
        okd = shim.shim_dantzig(n, _p(A), _p(xd), _p(b), _p(lo), _p(hi), fi.ctypes.data_as(pi))
        if okr == okd:
            agree += 1
            if okr == 1 and np.all(np.isfinite(xr)):
                both += 1
                assert np.allclose(xr, xd, rtol=1e-6, atol=1e-9)
    assert agree / N > 0.95 and both > 0.5 * N


def test_device_pgs_and_reduce_equal_the_oracle_restatement(shim):
    rng = np.random.default_rng(2)
    for trial in range(200):
        nc = int(rng.integers(1, 9)); n = 3 * nc
        A, b, lo, hi, fi = contact_lcp(rng, nc, int(rng.integers(3, 24)), cfm=1e-4)
        if trial % 3 == 0 and nc >= 2:      # duplicate a contact so that reduce() has something to merge
            A[3:6, :] = A[0:3, :]; A[:, 3:6] = A[:, 0:3]; b[3:6] = b[0:3]; hi[3:6] = hi[0:3]; lo[3:6] = lo[0:3]
        x0 = rng.normal(0, 0.1, n)
        xo = x0.copy(); xd = x0.copy()
        oko = L.nbo_lcp_pgs(n, _p(A), _p(xo), _p(b), _p(lo), _p(hi), fi.ctypes.data_as(pi), 30, C.c_double(1e-6), C.c_double(1e-3), C.c_double(1e-9))
        okd = shim.shim_pgs(n, _p(A), _p(xd), _p(b), _p(lo), _p(hi), fi.ctypes.data_as(pi))
        assert oko == okd and np.allclose(xo, xd, rtol=1e-12, atol=1e-14)
        for rf in (0, 1):
            Ar = np.zeros(n * n); xr = np.zeros(n); br = np.zeros(n); lor = np.zeros(n); hir = np.zeros(n); fr = np.zeros(n, np.int32); mo = np.zeros(n * n)
            nr = L.nbo_lcp_reduce(n, _p(A), _p(x0), _p(b), _p(lo), _p(hi), fi.ctypes.data_as(pi), rf, _p(Ar), _p(xr), _p(br), _p(lor), _p(hir), fr.ctypes.data_as(pi), _p(mo))
            Ad = np.zeros(n * n); xdv = np.zeros(n); bd = np.zeros(n); lod = np.zeros(n); hid = np.zeros(n); fd = np.zeros(n, np.int32); mt = np.zeros(n, np.int32)
            nd = shim.shim_reduce(n, _p(A), _p(x0), _p(b), _p(lo), _p(hi), fi.ctypes.data_as(pi), rf, _p(Ad), _p(xdv), _p(bd), _p(lod), _p(hid), fd.ctypes.data_as(pi), mt.ctypes.data_as(pi))
            assert nr == nd
            assert np.allclose(Ar[:nr * nr], Ad[:nr * nr]) and np.allclose(br[:nr], bd[:nr]) and np.array_equal(fr[:nr], fd[:nr])
            M = mo[:n * nr].reshape(n, nr)
            for o in range(n):
                assert (M[o].sum() == 0 and mt[o] == -1) or (M[o, mt[o]] == 1 and M[o].sum() == 1)


def test_device_cod_solves_equal_pinv(shim):
    rng = np.random.default_rng(3)
    for (c, k) in ((6, 6), (8, 5), (12, 6), (24, 12), (24, 24), (3, 1)):
        U = rng.normal(0, 1, (c, k)); A = _d(U @ U.T)                       # symmetric PSD like the clamping block of A
        if k == c:
            A = _d(rng.normal(0, 1, (c, c)))                                 # also a non-symmetric full-rank case
        b = _d(rng.normal(0, 1, c))
        for tr in (0, 1):
            x = np.zeros(c)
            rank = shim.shim_cod_solve(c, _p(A), _p(b), _p(x), tr)
            assert rank == k
            ref = np.linalg.pinv(A.T if tr else A, rcond=1e-11) @ b
            assert np.allclose(x, ref, rtol=1e-8, atol=1e-9)
