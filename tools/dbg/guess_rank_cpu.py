"""CPU: how often does the device's LCP code (host emulation, tests/host_shim) end on another solver stage than the oracle on the
oracle's own A, b - per soak mode - and is the first guess (pseudo-inverse route) the reason?  Worlds whose constraints are frictional
contacts of one constrained group only.   usage: python tools/dbg/guess_rank_cpu.py <stress mode|-> [first seed] [models] [B]"""
import ctypes as C, os, sys, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
src = open(os.path.join(ROOT, "tools", "soak_parity.py")).read().replace("import torch  # noqa: E402", "").replace(
    "from nimblephysics_amd.timestep import timestep  # noqa: E402", "")
mod = types.ModuleType("soak_cpu"); mod.__file__ = os.path.join(ROOT, "tools", "soak_parity.py"); exec(compile(src, "soak_cpu", "exec"), mod.__dict__)
import soak_stress, oracle
from oracle import OracleWorld
pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
_p = lambda a: a.ctypes.data_as(pd)
_pi = lambda a: a.ctypes.data_as(pi)
mode = sys.argv[1] if len(sys.argv) > 1 else "-"; first = int(sys.argv[2]) if len(sys.argv) > 2 else 43000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40; B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
shim = C.CDLL(os.path.join(ROOT, "tests", "host_shim", "libcoop_shim.so"))
OL = oracle._lib(); OL.nbo_lcp_guess.argtypes = [C.c_int, pd, pd, pi, pd]
tot = {"worlds": 0, "same_stage": 0, "other_stage": 0, "guess_differs_1e-6": 0, "other_stage_and_guess_differs": 0, "cond>1e10": 0, "other_stage_and_cond>1e10": 0}
for seed in range(first, first + count):
    case = mod.make_case(seed, B, False, False, True, False)
    if case is None:
        continue
    md, s, a, g = case if mode == "-" else soak_stress.mutator(mode)(seed, *case)
    ow = OracleWorld(md)
    for b in range(B):
        ow.reset_lcp_cache(); ow.step(s[b], a[b]); st = ow.last_status
        if not (st & 1) or (st & 0x482):
            continue
        L = ow.last_lcp(); m = len(L["b"]); fi = L["findex"]
        if m % 3 or m > 24 or any(fi[3 * c + 1] != 3 * c or fi[3 * c + 2] != 3 * c for c in range(m // 3)):
            continue
        A = L["A"] - (md.fallback_cfm * np.eye(m) if st & 0x18 else 0)
        nc = m // 3
        grp = list(range(nc))
        for i in range(nc):
            for j in range(nc):
                if np.abs(A[3 * i:3 * i + 3, 3 * j:3 * j + 3]).max() > 0:
                    gi, gj = grp[i], grp[j]; grp = [gi if x == gj else x for x in grp]
        if len(set(grp)) != 1:
            continue
        A24 = np.zeros((24, 24)); A24[:m, :m] = A; b24 = np.zeros(24); b24[:m] = L["b"]
        mu = np.ones(8); mu[:nc] = [L["hi"][3 * c + 1] for c in range(nc)]
        X = np.zeros(24); X0 = np.zeros(24); cls = np.zeros(24, np.int32); E = np.zeros(24)
        mask = (1 << m) - 1
        shim.shim_coop_stage0_masked(m, _p(np.ascontiguousarray(A24)), _p(b24), _p(mu), C.c_uint64(mask), _p(X), _p(X0), _pi(cls), _p(E))
        Xc = np.zeros(24); Xs = np.zeros(24); cls2 = np.zeros(24, np.int32); cfm = C.c_double(0)
        sd = shim.shim_coop_cascade_masked(m, _p(np.ascontiguousarray(A24)), _p(b24), _p(mu), _p(X0), C.c_uint64(mask), C.c_double(md.fallback_cfm), _p(Xc), C.byref(cfm), _p(Xs), _pi(cls2))
        xg = np.zeros(m); OL.nbo_lcp_guess(m, _p(np.ascontiguousarray(A)), _p(L["b"].copy()), _pi(fi.astype(np.int32)), _p(xg))
        gd = np.abs(X0[:m] - xg).max() > 1e-6 * max(np.abs(xg).max(), 1e-30)
        sv = np.linalg.svd(A, compute_uv=False); nz = sv[sv > 1e-13 * sv[0]]; cond = nz[0] / nz[-1]
        other = (sd & 0x3C) != (st & 0x3C)
        tot["worlds"] += 1; tot["same_stage"] += int(not other); tot["other_stage"] += int(other); tot["guess_differs_1e-6"] += int(gd)
        tot["other_stage_and_guess_differs"] += int(other and gd); tot["cond>1e10"] += int(cond > 1e10); tot["other_stage_and_cond>1e10"] += int(other and cond > 1e10)
print(mode, tot)
