"""Debug (GPU box): the guess matrix of one soak world through both device pseudo-inverse routes."""
import os, sys, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import soak_parity
from oracle import OracleWorld
from nimblephysics_amd._lib import check, lib
seed, mode, wd = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
md, s, a, g = soak_parity.make_case(seed, 256, big=mode == "big", multi=mode == "multi", balls=mode == "balls")
ow = OracleWorld(md); ow.reset_lcp_cache(); ow.step(s[wd], a[wd])
l = ow.last_lcp(); A = l["A"]; b = l["b"]; n = len(b)
Q = np.zeros((1, 24, 24)); Q[0, :n, :n] = A
size = np.array([n], np.int32)
vp = lambda x: C.c_void_p(x.ctypes.data)
for route in (0, 1):
    P = np.zeros_like(Q); r = np.zeros(1, np.int32)
    check(lib().nbl_selftest_pinv(1, vp(Q), vp(size), route, vp(P), vp(r), 1, None), "pinv")
    x = P[0, :n, :n] @ b
    print("route", route, "rank", r[0], "x0 =", np.array2string(x, precision=6), "err vs inv", np.abs(P[0, :n, :n] - np.linalg.inv(A)).max() / np.abs(np.linalg.inv(A)).max())
