"""Debug: the PRODUCT's device cascade (host emulation, tests/host_shim) on the LCP of one soak world as the oracle built it, group by group.
usage: python tools/dbg/cascade_world_dbg.py <seed> <world> [big|multi|balls] [stress mode of tools/soak_stress.py]"""
import ctypes as C
import os, sys, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
src = open(os.path.join(ROOT, "tools", "soak_parity.py")).read().replace("import torch  # noqa: E402", "").replace(
    "from nimblephysics_amd.timestep import timestep  # noqa: E402", "")
mod = types.ModuleType("soak_cpu"); mod.__file__ = os.path.join(ROOT, "tools", "soak_parity.py"); exec(compile(src, "soak_cpu", "exec"), mod.__dict__)
from oracle import OracleWorld
pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
_p = lambda a: a.ctypes.data_as(pd)
_pi = lambda a: a.ctypes.data_as(pi)
seed, wd = int(sys.argv[1]), int(sys.argv[2]); mode = sys.argv[3] if len(sys.argv) > 3 else ""
md, s, a, g = mod.make_case(seed, 256, big=mode == "big", multi=mode == "multi", balls=mode == "balls")
if len(sys.argv) > 4:
    import soak_stress
    md, s, a, g = soak_stress.mutator(sys.argv[4])(seed, md, s, a, g)
ow = OracleWorld(md)
ow.step(s[wd], a[wd])
L = ow.last_lcp()
m = len(L["b"]); nc = m // 3
print("oracle status", hex(ow.last_status), "rows", m, "row classes", L["row_class"])
if ow.last_status & 0x18:      # the oracle's record holds A with the fallback CFM on the diagonal once a CFM stage was used
    L["A"] = L["A"] - md.fallback_cfm * np.eye(m)
A = np.zeros((24, 24)); A[:m, :m] = L["A"]; b = np.zeros(24); b[:m] = L["b"]
mu = np.ones(8)
for c in range(nc):
    mu[c] = L["hi"][3 * c + 1] if L["findex"][3 * c + 1] >= 0 else 0.0
print("mu", mu[:nc], "findex", L["findex"])
# groups: connected components of the contact blocks of A
grp = list(range(nc))
for i in range(nc):
    for j in range(nc):
        if np.abs(A[3 * i:3 * i + 3, 3 * j:3 * j + 3]).max() > 0:
            gi, gj = grp[i], grp[j]
            grp = [gi if x == gj else x for x in grp]
print("contact groups", grp)
shim = C.CDLL(os.path.join(ROOT, "tests", "host_shim", "libcoop_shim.so"))
for gid in sorted(set(grp)):
    mask = 0
    for c in range(nc):
        if grp[c] == gid:
            mask |= 7 << (3 * c)
    X = np.zeros(24); X0 = np.zeros(24); cls = np.zeros(24, np.int32); E = np.zeros(24)
    ret = shim.shim_coop_stage0_masked(m, _p(np.ascontiguousarray(A)), _p(b), _p(mu), C.c_uint64(mask), _p(X), _p(X0), _pi(cls), _p(E))
    print(f"group {gid} mask {mask:#x}: stage0 ok {ret & 1}")
    if not (ret & 1):
        Xc = np.zeros(24); Xs = np.zeros(24); cls2 = np.zeros(24, np.int32); cfm = C.c_double(0)
        st = shim.shim_coop_cascade_masked(m, _p(np.ascontiguousarray(A)), _p(b), _p(mu), _p(X0), C.c_uint64(mask), C.c_double(md.fallback_cfm), _p(Xc), C.byref(cfm),
                                           _p(Xs), _pi(cls2))
        print(f"   device cascade status {st:#x} cfm {cfm.value}  x {Xc[:m]}")
print("oracle x", L["x"])

# ---- stage by stage on the plain arrays: oracle restatement vs device emulation ----
import oracle
OL = oracle._lib()
n = m
A6 = np.ascontiguousarray(L["A"]); lo = L["lo"].copy(); hi = L["hi"].copy(); fi = L["findex"].astype(np.int32)
x0 = X0[:n].copy()
print("guess x0", x0)
A2 = A6 + md.fallback_cfm * np.eye(n)
for rf in (0,):
    Ar = np.zeros(n * n); xr = np.zeros(n); br = np.zeros(n); lor = np.zeros(n); hir = np.zeros(n); fr = np.zeros(n, np.int32); mo = np.zeros(n * n)
    nr = OL.nbo_lcp_reduce(n, _p(np.ascontiguousarray(A2)), _p(x0), _p(L["b"].copy()), _p(lo), _p(hi), _pi(fi), rf, _p(Ar), _p(xr), _p(br), _p(lor), _p(hir), _pi(fr), _p(mo))
    print("oracle reduce ->", nr, "rows; findex", fr[:nr])
    xo = xr[:nr].copy()
    oko = OL.nbo_lcp_pgs(nr, _p(np.ascontiguousarray(Ar[:nr * nr])), _p(xo), _p(br[:nr].copy()), _p(lor[:nr].copy()), _p(hir[:nr].copy()), _pi(fr[:nr].copy()), 30, C.c_double(1e-6), C.c_double(1e-3), C.c_double(1e-9))
    print("oracle pgs ok", oko, "x", xo)
    A24 = np.zeros((24, 24)); A24[:nr, :nr] = Ar[:nr * nr].reshape(nr, nr)
    xd = np.zeros(24); xd[:nr] = xr[:nr]; b24 = np.zeros(24); b24[:nr] = br[:nr]; lo24 = np.zeros(24); lo24[:nr] = lor[:nr]; hi24 = np.zeros(24); hi24[:nr] = hir[:nr]
    f24 = np.full(24, -1, np.int32); f24[:nr] = fr[:nr]
    okd = shim.shim_coop_pgs(nr, _p(np.ascontiguousarray(A24[:nr, :nr])), _p(xd), _p(b24), _p(lo24), _p(hi24), _pi(f24))
    print("device pgs ok", okd, "x", xd[:nr])
    Ad = np.zeros(n * n); xdv = np.zeros(n); bd = np.zeros(n); lod = np.zeros(n); hid = np.zeros(n); fd = np.zeros(n, np.int32); mt = np.zeros(n, np.int32)
    nd = shim.shim_coop_reduce(n, _p(np.ascontiguousarray(A2)), _p(x0), _p(L["b"].copy()), _p(lo), _p(hi), _pi(fi), rf, _p(Ad), _p(xdv), _p(bd), _p(lod), _p(hid), _pi(fd), _pi(mt))
    print("device reduce ->", nd, "rows; findex", fd[:nd], "mapTo", mt, "max |A diff|", np.abs(Ad[:nd*nd] - Ar[:nr*nr]).max() if nd == nr else None)
    print("lo", lod[:nd], lor[:nr]); print("hi", hid[:nd], hir[:nr]); print("x", xdv[:nd], xr[:nr])
# validity of the PGS solution on the CFM matrix
X = np.zeros(n); X[:] = xo
v = A2 @ X - L["b"]
print("v = (A + cfm) x - b:", v)
print("bounds friction rows:", [(X[i], hi[i] * X[fi[i]]) for i in range(n) if fi[i] >= 0])
cl = [i for i in range(n) if fi[i] >= 0 or L["b"][i] > 0]
xg = np.zeros(n); xg[cl] = np.linalg.pinv(A6[np.ix_(cl, cl)], rcond=1e-12) @ L["b"][cl]
print("numpy guess", xg); print("device X0  ", x0); print("b", L["b"]); print("sv of A_cl", np.linalg.svd(A6[np.ix_(cl, cl)], compute_uv=False))

# ---- sensitivity of the device cascade (emulation) to the last bits of A ----
import collections
rng = np.random.default_rng(0)
mu8 = mu
for scale in (1e-16, 1e-15, 1e-14, 1e-13):
    cnt = collections.Counter()
    for t in range(200):
        E_ = rng.normal(0, scale, (m, m)); E_ = (E_ + E_.T) / 2
        A24 = np.zeros((24, 24)); A24[:m, :m] = L["A"] * (1.0 + E_ / np.maximum(np.abs(L["A"]), 1e-300) * 0) + E_
        X = np.zeros(24); X0_ = np.zeros(24); cls = np.zeros(24, np.int32); E2 = np.zeros(24)
        ret = shim.shim_coop_stage0_masked(m, _p(np.ascontiguousarray(A24)), _p(b), _p(mu8), C.c_uint64((1 << m) - 1), _p(X), _p(X0_), _pi(cls), _p(E2))
        Xc = np.zeros(24); Xs = np.zeros(24); cls2 = np.zeros(24, np.int32); cfm = C.c_double(0)
        stt = shim.shim_coop_cascade_masked(m, _p(np.ascontiguousarray(A24)), _p(b), _p(mu8), _p(X0_), C.c_uint64((1 << m) - 1), C.c_double(md.fallback_cfm), _p(Xc), C.byref(cfm), _p(Xs), _pi(cls2))
        cnt[(ret & 1, hex(stt))] += 1
    print(f"|dA| ~ {scale:g}: {dict(cnt)}")
