"""CPU: the oracle's LCP of one mixed-soak world through the oracle's own stage functions, as is and with the upper-limit rows negated
(the device's form).  usage: python tools/dbg/mix_lcp_cpu.py <seed> <world> <variant> [mode]"""
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
seed, wd, variant = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
mode = sys.argv[4] if len(sys.argv) > 4 else "mix"
md, s, a, g = mod.make_case(seed, 256, variant == "big", variant == "multi", variant == "balls", False)
md, s, a, g = soak_stress.mutator(mode)(seed, md, s, a, g)
ow = OracleWorld(md); ow.step(s[wd], a[wd]); L = ow.last_lcp()
np.set_printoptions(linewidth=220, precision=10)
m = len(L["b"]); print("status", hex(ow.last_status), "rows", m, "findex", L["findex"], "classes", L["row_class"])
A = L["A"].copy()
if ow.last_status & 0x18:
    A = A - md.fallback_cfm * np.eye(m)
print("A\n", A); print("b", L["b"]); print("lo", L["lo"]); print("hi", L["hi"]); print("x", L["x"])
OL = oracle._lib()
OL.nbo_lcp_dantzig.argtypes = [C.c_int, pd, pd, pd, pd, pd, pi, C.c_int]
OL.nbo_lcp_valid.argtypes = [C.c_int, pd, pd, pd, pd, pd, pi, C.c_int]
OL.nbo_lcp_guess.argtypes = [C.c_int, pd, pd, pi, pd]
for neg in (False, True):
    sg = np.ones(m)
    if neg:
        sg[np.isinf(L["lo"]) & (L["hi"] == 0)] = -1.0
    A2 = np.ascontiguousarray(A * sg[:, None] * sg[None, :]); b2 = L["b"] * sg
    lo2 = np.where(sg < 0, -L["hi"], L["lo"]); hi2 = np.where(sg < 0, -L["lo"], L["hi"])
    fi = L["findex"].astype(np.int32)
    x0 = np.zeros(m); OL.nbo_lcp_guess(m, _p(A2), _p(b2), _pi(fi), _p(x0))
    v0 = OL.nbo_lcp_valid(m, _p(A2), _p(x0), _p(b2), _p(lo2), _p(hi2), _pi(fi), 0)
    x = x0.copy(); ok = OL.nbo_lcp_dantzig(m, _p(A2), _p(x), _p(b2), _p(lo2), _p(hi2), _pi(fi), 0)
    v = OL.nbo_lcp_valid(m, _p(A2), _p(x), _p(b2), _p(lo2), _p(hi2), _pi(fi), 0)
    print("negated" if neg else "as is  ", "guess", x0 * sg, "valid", v0, "| dantzig ok", ok, "x", x * sg, "valid", v, "w", (A2 @ x - b2) * sg)
OL.nbo_lcp_cascade.argtypes = [C.c_int, pd, pd, pd, pd, pd, pi, C.c_double, pd, C.POINTER(C.c_uint32), pd]
for neg in (False, True):
    sg = np.ones(m)
    if neg:
        sg[np.isinf(L["lo"]) & (L["hi"] == 0)] = -1.0
    A2 = np.ascontiguousarray(A * sg[:, None] * sg[None, :]); b2 = L["b"] * sg
    lo2 = np.where(sg < 0, -L["hi"], L["lo"]); hi2 = np.where(sg < 0, -L["lo"], L["hi"])
    fi = L["findex"].astype(np.int32)
    x0 = np.zeros(m); OL.nbo_lcp_guess(m, _p(A), _p(L["b"].copy()), _pi(fi), _p(x0)); x0 = x0 * sg      # the reference's guess, carried over
    xo = np.zeros(m); st = C.c_uint32(0); cfm = C.c_double(0)
    print("--- cascade", "negated" if neg else "as is")
    sys.stdout.flush()
    ok = OL.nbo_lcp_cascade(m, _p(A2), _p(x0), _p(b2), _p(lo2), _p(hi2), _pi(fi), C.c_double(md.fallback_cfm), _p(xo), C.byref(st), C.byref(cfm))
    print("   valid", ok, "status", hex(st.value), "cfm", cfm.value, "x", xo * sg)

# ---- the device's code (host emulation) on the same LCP in the device's layout: one 3-row slot per constraint, frictionless contacts and
#      joint-limit rows on the slot's first row, upper-limit rows negated ----
shim = C.CDLL(os.path.join(ROOT, "tests", "host_shim", "libcoop_shim.so"))
fi = L["findex"]; rows = []; slot = 0; limMask = negMask = 0; mu = np.zeros(8); r = 0
dev_of = np.zeros(m, int); sgn = np.ones(m)
while r < m:
    if r + 2 < m and fi[r + 1] == r and fi[r + 2] == r:          # contact with friction
        dev_of[r:r + 3] = [3 * slot, 3 * slot + 1, 3 * slot + 2]; mu[slot] = L["hi"][r + 1]; r += 3
    else:
        dev_of[r] = 3 * slot
        if np.isinf(L["lo"][r]) or (L["hi"][r] == 0):           # upper-limit row
            limMask |= 1 << (3 * slot); negMask |= 1 << (3 * slot); sgn[r] = -1.0
        elif r >= 3 * len(ow.last_contacts()) - 2 * sum(1 for c in range(0)):   # (lower-limit rows are told from frictionless contacts below)
            pass
        r += 1
    slot += 1
# lower-limit rows: the rows after the last contact's rows
nct = len(ow.last_contacts()); crow = 0
for c in range(nct):
    crow += 3 if (crow + 2 < m and fi[crow + 1] == crow) else 1
for r in range(crow, m):
    limMask |= 1 << dev_of[r]
md_rows = 3 * slot
A24 = np.zeros((24, 24)); b24 = np.zeros(24)
for i in range(m):
    b24[dev_of[i]] = L["b"][i] * sgn[i]
    for j in range(m):
        A24[dev_of[i], dev_of[j]] = A[i, j] * sgn[i] * sgn[j]
print("device layout: rows", md_rows, "limMask", hex(limMask), "negMask", hex(negMask), "mu", mu[:slot])
X = np.zeros(24); X0 = np.zeros(24); cls = np.zeros(24, np.int32); E = np.zeros(24)
mask = (1 << md_rows) - 1
ret = shim.shim_coop_stage0_lim(md_rows, _p(np.ascontiguousarray(A24)), _p(b24), _p(mu), C.c_uint(mask), C.c_uint(limMask), C.c_uint(negMask), _p(X), _p(X0), _pi(cls), _p(E))
print("emulation stage0 ok", ret & 1, "X0", X0[:md_rows])
Xc = np.zeros(24); cfm = C.c_double(0); stg = np.zeros(3, np.int32)
st = shim.shim_coop_cascade_lim(md_rows, _p(np.ascontiguousarray(A24)), _p(b24), _p(mu), _p(X0), C.c_uint(mask), C.c_uint(limMask), C.c_uint(negMask),
                                C.c_double(md.fallback_cfm), _p(Xc), C.byref(cfm), _pi(stg))
print("emulation cascade status", hex(st), "stage flags (1 solved, 2 valid, 4 nan)", stg, "cfm", cfm.value, "x", Xc[:md_rows])

# ---- sensitivity of the ORACLE's cascade to one-ulp noise on b, on A, on both ----
if os.environ.get("NOISE"):
    import collections
    rng = np.random.default_rng(0)
    fi = L["findex"].astype(np.int32)
    x0 = np.zeros(m); OL.nbo_lcp_guess(m, _p(A), _p(L["b"].copy()), _pi(fi), _p(x0))
    for what in ("b", "A", "A+b", "Q", "N"):
        for ulps in ((1, 4) if what != "Q" else (1, 4, 16, 64)):
            cnt = collections.Counter()
            for t in range(200):
                A2 = A.copy(); b2 = L["b"].copy()
                if "A" in what:
                    N = np.triu(rng.integers(-1, 2, (m, m))); N = N + np.triu(N, 1).T
                    A2 = A * (1 + N * ulps * 2.220446049250313e-16)
                if what == "Q":                                   # absolute: ulps * eps * max |A| on every entry, symmetric
                    N = np.triu(rng.integers(-1, 2, (m, m))); N = N + np.triu(N, 1).T
                    A2 = A + N * ulps * 2.220446049250313e-16 * np.abs(A).max()
                if what == "N":                                   # like A, but (i, j) and (j, i) drawn independently
                    A2 = A * (1 + rng.integers(-1, 2, (m, m)) * ulps * 2.220446049250313e-16)
                if "b" in what:
                    b2 = b2 * (1 + rng.integers(-1, 2, m) * ulps * 2.220446049250313e-16)
                x0 = np.zeros(m); OL.nbo_lcp_guess(m, _p(np.ascontiguousarray(A2)), _p(b2), _pi(fi), _p(x0))
                xo = np.zeros(m); st = C.c_uint32(0); cfm = C.c_double(0)
                ok = OL.nbo_lcp_cascade(m, _p(np.ascontiguousarray(A2)), _p(x0), _p(b2), _p(L["lo"].copy()), _p(L["hi"].copy()), _pi(fi), C.c_double(md.fallback_cfm), _p(xo), C.byref(st), C.byref(cfm))
                cnt[(hex(st.value), tuple(np.round(xo, 3)))] += 1
            print(what, ulps, "ulps:", [(k[0], k[1][0], k[1][-1], v) for k, v in cnt.items()])
