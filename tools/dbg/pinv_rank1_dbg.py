"""Rank decisions of the device pseudo-inverses on NON-symmetric low-rank matrices (the folded Q of a friction row on its bound over a
rank-1 Delassus matrix: contacts between two bodies joined by one revolute joint)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from nimblephysics_amd._lib import check, lib
rng = np.random.default_rng(5)
count = 600
Q = np.zeros((count, 24, 24)); size = np.zeros(count, np.int32); rank = np.zeros(count, np.int32)
for t in range(count):
    c = int(rng.integers(3, 13)); k = int(rng.integers(1, 4))
    U = rng.normal(0, 1, (c, k)) * 10 ** rng.uniform(-2, 1); V = rng.normal(0, 1, (c, k))
    sub = U @ V.T if t % 2 else U @ (U + 0.3 * V).T
    Q[t, :c, :c] = sub; size[t] = c; rank[t] = k
vp = lambda a: C.c_void_p(a.ctypes.data)
P = np.zeros_like(Q); r = np.zeros(count, np.int32)
check(lib().nbl_selftest_pinv(count, vp(Q), vp(size), 0, vp(P), vp(r), 1, None), "nbl_selftest_pinv")
bad = np.where(r != rank)[0]
print("rank mismatches", len(bad), "of", count, [(int(size[t]), int(rank[t]), int(r[t])) for t in bad[:12]])
worst = 0
for t in range(count):
    if r[t] == rank[t]:
        ref = np.linalg.pinv(Q[t], rcond=1e-10); worst = max(worst, np.abs(P[t] - ref).max() / np.abs(ref).max())
print("worst rel err when rank right", worst)
