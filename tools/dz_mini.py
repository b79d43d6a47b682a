"""Developer loop for the cooperative Dantzig driver: tools/dz_mini.hip built as tools/dbg/libdz_mini*.so (seconds), run on the REDUCED stage-1
problems of the metric distribution (exactly what k_contact_cascade_stages hands to coopDantzig: the oracle's A, b, bounds after LCPUtils::reduce
is not applied here - the unreduced 24-row problems - plus early-terminating ones), checked bit for bit against the reference's own
dSolveLCP (oracle/_ref).  usage (GPU box): python tools/dz_mini.py lib1.so [lib2.so ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401,E402  (the HIP runtime of the process)
import oracle  # noqa: E402

from oracle import OracleWorld  # noqa: E402
from util import contact_inputs  # noqa: E402

n = 24
OL = oracle._lib()
pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)


def stage1_problems(nworlds=2048, seed=1000):
    """What coopCascadeStage1 hands to coopDantzig on the metric distribution: the world's LCP (without the fallback CFM the record carries
    once a CFM stage ran) after LCPUtils::reduce, padded to 24 rows."""
    md, s, a = contact_inputs("atlas20", nworlds, seed, joint_noise=0.02, vel_noise=0.01, action_noise=0.1)
    ow = OracleWorld(md)
    out = []
    for i in range(nworlds):
        ow.reset_lcp_cache(); ow.step(s[i], a[i])
        if ow.last_status & 0x2:
            continue
        l = ow.last_lcp()
        m = len(l["b"])
        if m != 24:
            continue
        A = l["A"] - (md.fallback_cfm * np.eye(m) if (ow.last_status & 0x18) else 0.0)
        fi = l["findex"].astype(np.int32)
        lo = np.where(fi >= 0, -np.abs(l["hi"]), 0.0); hi = np.where(fi >= 0, np.abs(l["hi"]), np.inf)
        x0 = np.zeros(m)
        Ar = np.zeros(m * m); xr = np.zeros(m); br = np.zeros(m); lor = np.zeros(m); hir = np.zeros(m); fr = np.zeros(m, np.int32); mo = np.zeros(m * m)
        _p = lambda v: v.ctypes.data_as(pd)
        nr = OL.nbo_lcp_reduce(m, _p(np.ascontiguousarray(A)), _p(x0), _p(l["b"].copy()), _p(lo.copy()), _p(hi.copy()), fi.copy().ctypes.data_as(pi), 0, _p(Ar), _p(xr), _p(br), _p(lor), _p(hir),
                               fr.ctypes.data_as(pi), _p(mo))
        Ap = np.zeros((24, 24)); Ap[:nr, :nr] = Ar[:nr * nr].reshape(nr, nr)
        pad = lambda v, fill=0.0: np.concatenate([v[:nr], np.full(24 - nr, fill)])
        out.append((nr, Ap, pad(br), pad(lor), pad(hir), np.concatenate([fr[:nr], np.full(24 - nr, -1, np.int32)]).astype(np.int32)))
    return out


cache = os.path.join(ROOT, "tools", "dbg", "dz_problems.npz")
if os.path.exists(cache):
    z = np.load(cache); ns, A, b, lo, hi, fi = z["ns"], z["A"], z["b"], z["lo"], z["hi"], z["fi"]
else:
    P = stage1_problems()
    ns = np.array([p[0] for p in P], np.int32)
    A = np.stack([p[1] for p in P]); b = np.stack([p[2] for p in P]); lo = np.stack([p[3] for p in P]); hi = np.stack([p[4] for p in P]); fi = np.stack([p[5] for p in P])
    np.savez(cache, ns=ns, A=A, b=b, lo=lo, hi=hi, fi=fi)
ref = []
for k in range(len(A)):
    m = int(ns[k]); xr = np.zeros(m)
    Ak = np.ascontiguousarray(A[k][:m, :m])
    okr = OL.nbo_lcp_dantzig(m, Ak.ctypes.data_as(pd), xr.ctypes.data_as(pd), b[k][:m].copy().ctypes.data_as(pd), lo[k][:m].copy().ctypes.data_as(pd),
                             hi[k][:m].copy().ctypes.data_as(pd), fi[k][:m].copy().ctypes.data_as(pi), 1)
    ref.append((okr, xr))
print(f"{len(A)} stage-1 problems; rows: mean {ns.mean():.1f}; the reference solves {np.mean([r[0] == 1 for r in ref]):.2f}, exits early on {np.mean([r[0] == 0 for r in ref]):.2f}")
vp = lambda a_: C.c_void_p(a_.ctypes.data)
for path in sys.argv[1:]:
    L = C.CDLL(os.path.abspath(path))
    out = []
    for count, reps in ((370, 20), (8192, 3)):
        idx = np.arange(count) % len(A)
        n_, a_, b_, lo_, hi_, fi_ = (np.ascontiguousarray(v[idx]) for v in (ns, A, b, lo, hi, fi))
        x = np.zeros((count, n)); rc = np.zeros(count, np.int32); ms = C.c_double(0)
        assert L.dz_run(count, n, vp(n_), vp(a_), vp(b_), vp(lo_), vp(hi_), vp(fi_), vp(x), vp(rc), reps, C.byref(ms)) == 0
        bad = sum(1 for k in range(min(count, len(A))) if rc[k] != -1 and (ref[k][0] != rc[k] or (ref[k][0] == 1 and not np.array_equal(ref[k][1], x[k][:ns[k]]))))
        out.append(f"count {count}: {ms.value * 1e3:7.1f} us ({ms.value * 1e3 / count * 1024:6.1f} us / 1024)  not bit-identical: {bad}")
    print(os.path.basename(path), "|", " | ".join(out), flush=True)
