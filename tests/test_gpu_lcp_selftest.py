"""The device Dantzig driver (stage 1 of the LCP cascade, coop_dantzig_dev.hpp) run ON THE GPU through the C ABI
(nbl_selftest_lcp_dantzig) against the reference's own dSolveLCP (oracle/_ref, compiled from
dart/external/odelcpsolver): success flag and every bit of x identical, rank-deficient problems included."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problems(rng, count, nc, ndof):
    from util import contact_lcp
    n = 3 * nc
    A = np.zeros((count, n, n)); b = np.zeros((count, n)); lo = np.zeros((count, n)); hi = np.zeros((count, n)); fi = np.zeros((count, n), np.int32)
    for i in range(count):
        A[i], b[i], lo[i], hi[i], fi[i] = contact_lcp(rng, nc, ndof)
    return A, b, lo, hi, fi


# (more than 8 contacts: the code of the 48-row instantiation of the library, csrc/abi_variants.h)
@pytest.mark.parametrize("nc,ndof", [(8, 30), (8, 12), (8, 6), (4, 6), (2, 3), (5, 6), (1, 6), (16, 60), (16, 12), (12, 18), (9, 6), (11, 40)])
def test_device_dantzig_bit_identical_to_reference(nc, ndof):
    import oracle
    from nimblephysics_amd._lib import check, lib
    if not os.path.exists(os.path.join(os.path.dirname(oracle.__file__), "_ref", "libodelcp_ref.so")):
        pytest.skip("oracle/_ref not built")
    OL = oracle._lib()
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    rng = np.random.default_rng(100 * nc + ndof)
    count, n = (512 if nc <= 8 else 192), 3 * nc
    A, b, lo, hi, fi = _problems(rng, count, nc, ndof)
    x = np.zeros((count, n)); rc = np.zeros(count, np.int32)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    check(lib().nbl_selftest_lcp_dantzig(count, n, vp(A), vp(b), vp(lo), vp(hi), vp(fi), vp(x), vp(rc)), "nbl_selftest_lcp_dantzig")
    solved = failed = 0
    for i in range(count):
        xr = np.zeros(n)
        okr = OL.nbo_lcp_dantzig(n, A[i].ctypes.data_as(pd), xr.ctypes.data_as(pd), b[i].copy().ctypes.data_as(pd), lo[i].copy().ctypes.data_as(pd),
                                 hi[i].copy().ctypes.data_as(pd), fi[i].copy().ctypes.data_as(pi), 1)
        if rc[i] == -1:
            assert okr == 0 or not np.all(np.isfinite(xr)), (i, okr)
            continue
        assert okr == rc[i], (i, okr, rc[i])
        if okr == 1:
            solved += 1
            assert np.array_equal(xr, x[i]), (i, np.abs(xr - x[i]).max())
        else:
            failed += 1
    assert solved > count // 4
    if ndof < 3 * nc:
        assert failed > 0          # singular A(C,C): the early exit is really exercised
