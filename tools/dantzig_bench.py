"""Micro-benchmark of the device Dantzig driver (stage 1 of the LCP cascade) on REAL problems: the 24-row LCPs of the metric
distribution (Atlas-20 on the ground, pose noise 0.02) that stage 0 does not resolve, as the oracle builds them (A, b, bounds, findex),
solved `count` at a time by nbl_selftest_lcp_dantzig_timed (one wavefront per problem, exactly the code of k_contact_cascade_stages'
stage 1 after `reduce`).  Prints the launch duration for one wave per SIMD (latency: the slowest problem sets it) and for a saturating
launch (throughput), checks x and the return codes bit for bit against the reference's own dSolveLCP (oracle/_ref).
usage (GPU box): python tools/dantzig_bench.py [worlds=1024] [alt .so]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from oracle import OracleWorld  # noqa: E402
from util import contact_inputs  # noqa: E402

if len(sys.argv) > 2:
    import nimblephysics_amd._lib as _l
    _l.LIB_PATH = os.path.abspath(sys.argv[2])
from nimblephysics_amd._lib import check, lib  # noqa: E402


def problems(nworlds, seed=1000):
    md, s, a = contact_inputs("atlas20", nworlds, seed, joint_noise=0.02, vel_noise=0.01, action_noise=0.1)
    ow = OracleWorld(md)
    out = []
    for i in range(nworlds):
        ow.reset_lcp_cache(); ow.step(s[i], a[i])
        if ow.last_status & 0x2:
            continue                                   # resolved at stage 0: never reaches the solver
        l = ow.last_lcp()
        n = len(l["b"])
        if n != 24:
            continue
        fi = l["findex"].astype(np.int32)
        lo = np.where(fi >= 0, -np.abs(l["hi"]), 0.0); hi = np.where(fi >= 0, np.abs(l["hi"]), np.inf)
        out.append((l["A"], l["b"], lo, hi, fi))
    return out


def main():
    nworlds = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    P = problems(nworlds)
    n = 24
    print(f"{len(P)} unresolved 24-row problems of {nworlds} worlds")
    OL = oracle._lib()
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    vp = lambda a_: C.c_void_p(a_.ctypes.data)
    for count, reps in ((min(len(P), 370), 20), (1024, 10), (8192, 3)):
        idx = np.arange(count) % len(P)
        A = np.ascontiguousarray(np.stack([P[i][0] for i in idx])); b = np.ascontiguousarray(np.stack([P[i][1] for i in idx]))
        lo = np.ascontiguousarray(np.stack([P[i][2] for i in idx])); hi = np.ascontiguousarray(np.stack([P[i][3] for i in idx]))
        fi = np.ascontiguousarray(np.stack([P[i][4] for i in idx]))
        x = np.zeros((count, n)); rc = np.zeros(count, np.int32); ms = C.c_double(0)
        check(lib().nbl_selftest_lcp_dantzig_timed(count, n, vp(A), vp(b), vp(lo), vp(hi), vp(fi), vp(x), vp(rc), reps, C.byref(ms)), "selftest")
        print(f"count {count:5d}: {ms.value * 1e3:8.1f} us per launch  ({ms.value * 1e3 / count * 1024:7.1f} us per 1024 problems); rc: solved {(rc == 1).mean():.2f} early exit {(rc == 0).mean():.2f} nan {(rc < 0).mean():.2f}")
    if os.path.exists(os.path.join(os.path.dirname(oracle.__file__), "_ref", "libodelcp_ref.so")):
        bad = 0
        for k in range(min(count, len(P))):
            xr = np.zeros(n)
            okr = OL.nbo_lcp_dantzig(n, A[k].ctypes.data_as(pd), xr.ctypes.data_as(pd), b[k].copy().ctypes.data_as(pd), lo[k].copy().ctypes.data_as(pd),
                                     hi[k].copy().ctypes.data_as(pd), fi[k].copy().ctypes.data_as(pi), 1)
            if rc[k] == -1:
                continue
            if okr != rc[k] or (okr == 1 and not np.array_equal(xr, x[k])):
                bad += 1
        print("problems not bit-identical to the reference's dSolveLCP:", bad)


if __name__ == "__main__":
    main()
