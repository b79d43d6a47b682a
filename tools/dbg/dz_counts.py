import os, sys, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["NBL_LIB_PATH"] = os.path.join(ROOT, "tools", "dbg", "libnimble_amd_counts.so")
import numpy as np
import dantzig_bench as db
from nimblephysics_amd._lib import check, lib
P = db.problems(1024)
n = 24; count = len(P)
A = np.ascontiguousarray(np.stack([p[0] for p in P])); b = np.ascontiguousarray(np.stack([p[1] for p in P])); lo = np.ascontiguousarray(np.stack([p[2] for p in P]))
hi = np.ascontiguousarray(np.stack([p[3] for p in P])); fi = np.ascontiguousarray(np.stack([p[4] for p in P]))
x = np.zeros((count, n)); rc = np.zeros(count, np.int32)
vp = lambda a_: C.c_void_p(a_.ctypes.data)
check(lib().nbl_selftest_lcp_dantzig(count, n, vp(A), vp(b), vp(lo), vp(hi), vp(fi), vp(x), vp(rc)), "selftest")
it, rem, tr = x[:, 0], x[:, 1], x[:, 2]
for nm, v in (("pivot iterations", it), ("C->N removals", rem), ("N->C transfers", tr)):
    print(f"{nm:18s} mean {v.mean():6.1f} p50 {np.percentile(v, 50):5.0f} p90 {np.percentile(v, 90):5.0f} p99 {np.percentile(v, 99):5.0f} max {v.max():5.0f}")
print("rc", np.bincount(rc + 1))
