"""The reference's own LCP fixtures (unittests/unit/test_LCPUtils.cpp:198-697, tests/golden/lcp_fixtures.json) through the WHOLE
solver cascade of BoxedLcpConstraintSolver::solveLcp (:461-687: reduce + Dantzig -> CFM + PGS -> friction dropped + PGS):
  (b) the oracle's cascade (stage 1 = the reference's own dSolveLCP, oracle/_ref) ends with a solution that passes
      LCPUtils::isLCPSolutionValid - the reference tests' pass criterion;
  (c) the PRODUCT's device cascade (the three stage functions + the order of preference that k_contact_cascade_stages / _final
      run, compiled for the host on the wave emulation) chooses the same stage and the same x as the oracle on these fixtures
      and on random rank-deficient contact problems.
A checker for device code, not a CPU path of the product."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from test_coop_host import shim  # noqa: F401  (fixture: builds tests/host_shim/libcoop_shim.so)
from util import contact_lcp, have_ref

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "lcp_fixtures.json")))
OL = oracle._lib()
pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
OL.nbo_lcp_cascade.argtypes = [C.c_int, pd, pd, pd, pd, pd, pi, C.c_double, pd, C.POINTER(C.c_uint32), pd]
CFM = 1e-4            # World.cpp:85 mFallbackConstraintForceMixingConstant
STAGE_BITS = 0x4 | 0x8 | 0x10 | 0x20 | 0x40


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(pd)


def _fixture(name):
    f = FIX[name]
    n = len(f["b"])
    return n, _d(f["A"]).reshape(n, n), _d(f.get("x", np.zeros(n))), _d(f["b"]), _d(f["lo"]), _d(f["hi"]), np.ascontiguousarray(f["fIndex"], dtype=np.int32)


def oracle_cascade(n, A, x0, b, lo, hi, fi, cfm=CFM):
    x = np.zeros(n); st = C.c_uint32(0); cfm_used = C.c_double(0)
    valid = OL.nbo_lcp_cascade(n, _p(_d(A)), _p(_d(x0)), _p(_d(b)), _p(_d(lo)), _p(_d(hi)), fi.ctypes.data_as(pi), cfm, _p(x), C.byref(st), C.byref(cfm_used))
    return x, st.value, cfm_used.value, bool(valid)


def device_cascade(shim, n, A, x0, b, lo, hi, fi, cfm=CFM):
    R, NC = shim.R, shim.NC            # both instantiations of the device code: 24 and 48 LCP rows
    assert n % 3 == 0 and n <= R
    A24 = np.zeros((R, R)); A24[:n, :n] = A
    b24 = np.zeros(R); b24[:n] = b
    x24 = np.zeros(R); x24[:n] = x0
    mu = np.zeros(NC); mu[:n // 3] = hi[1::3]
    assert np.all(fi[0::3] == -1) and np.all(fi[1::3] == np.arange(0, n, 3)) and np.all(fi[2::3] == np.arange(0, n, 3))
    X = np.zeros(R); cfm_used = C.c_double(0)
    shim.shim_coop_cascade.argtypes = [C.c_int, pd, pd, pd, pd, C.c_double, pd, pd]
    st = shim.shim_coop_cascade(n, _p(A24), _p(b24), _p(mu), _p(x24), cfm, _p(X), C.byref(cfm_used))
    return X[:n].copy(), st, cfm_used.value


# Fixtures for which the reference holds no validity criterion: REAL_LIFE_FAILURE_1 only checks reduce() == three manual merges
# (test_LCPUtils.cpp:423-463, restated in test_oracle_contact.py) and REAL_LIFE_FAILURE_3's test is commented out upstream
# (:465-556 sit inside a /* */ block).  With the production options (30 PGS sweeps, CFM 1e-4) no stage converges on them - they
# are the reference's "real life failures" - and the solver keeps the frictionless PGS iterate (BoxedLcpConstraintSolver.cpp:679-687).
NO_CRITERION = {"REAL_LIFE_FAILURE_1", "REAL_LIFE_FAILURE_3"}


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (reference Dantzig) not built")
@pytest.mark.parametrize("name", sorted(FIX))
@pytest.mark.parametrize("start", ["fixture_x", "zeros"])
def test_oracle_cascade_solves_the_reference_fixtures(name, start):
    """(b) the reference tests' criterion (LCPUtils::isLCPSolutionValid) on the result of the whole cascade, started from the
    fixture's own x and from zero as the pre-solve x."""
    n, A, x0, b, lo, hi, fi = _fixture(name)
    x, st, cfm_used, valid = oracle_cascade(n, A, x0 if start == "fixture_x" else np.zeros(n), b, lo, hi, fi)
    assert np.all(np.isfinite(x))
    assert st & (0x4 | 0x8 | 0x10), hex(st)                  # some stage produced the result
    if name in NO_CRITERION:
        assert st & 0x20 and not valid                       # every stage failed, like in the reference
    else:
        assert valid and not (st & 0x20), (name, hex(st), x)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (reference Dantzig) not built")
@pytest.mark.parametrize("name", sorted(FIX))
def test_device_cascade_equals_the_oracle_on_the_reference_fixtures(shim, name):
    """(c) same stage, same x (Dantzig results bit for bit, PGS results to round-off: the device sweeps in residual form)."""
    n, A, x0, b, lo, hi, fi = _fixture(name)
    xo, sto, cfmo, _ = oracle_cascade(n, A, x0, b, lo, hi, fi)
    xd, std, cfmd = device_cascade(shim, n, A, x0, b, lo, hi, fi)
    assert (std & STAGE_BITS) == (sto & STAGE_BITS), (name, hex(std), hex(sto))
    assert cfmd == cfmo
    if sto & 0x4:
        assert np.array_equal(xd, xo)
    else:
        assert np.allclose(xd, xo, rtol=1e-9, atol=1e-12)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (reference Dantzig) not built")
def test_device_cascade_equals_the_oracle_on_random_contact_problems(shim):
    """Rank-deficient (6- and 12-DOF) and full-rank problems of 1..8 frictional contacts from random pre-solve x: the device
    cascade takes the same path as the oracle's on EVERY problem (the Dantzig stage is bit-identical, so its early exits
    agree too) and returns the same x."""
    rng = np.random.default_rng(7)
    stages = {0x4: 0, 0x8: 0, 0x10: 0}
    for trial in range(18 if shim.R == 24 else 4):
        nc = int(rng.integers(1, shim.NC + 1)); n = 3 * nc
        ndof = int(rng.choice([6, 12, n + 3]))
        A, b, lo, hi, fi = contact_lcp(rng, nc, ndof)
        A = _d(np.tril(A) + np.tril(A, -1).T)                  # exactly symmetric, like the A the impulse tests build (earlier rows mirrored)
        if trial % 4 == 0:
            b = -np.abs(b)                                     # separating contacts: Dantzig solves with x = 0 on many rows
        x0 = rng.normal(0, 0.05, n) * (trial % 2)
        xo, sto, cfmo, _ = oracle_cascade(n, A, x0, b, lo, hi, fi)
        xd, std, cfmd = device_cascade(shim, n, A, x0, b, lo, hi, fi)
        assert (std & STAGE_BITS) == (sto & STAGE_BITS), (trial, n, ndof, hex(std), hex(sto))
        assert cfmd == cfmo
        if sto & 0x4:
            assert np.array_equal(xd, xo), (trial, np.abs(xd - xo).max())
        else:
            assert np.allclose(xd, xo, rtol=1e-8, atol=1e-11), (trial, hex(sto), np.abs(xd - xo).max())
        for bit in stages:
            stages[bit] += int(bool(sto & bit))
    assert stages[0x4] > (3 if shim.R == 24 else 0) and stages[0x10] > (3 if shim.R == 24 else 0), stages       # both ends of the cascade were exercised
