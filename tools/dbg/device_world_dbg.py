"""Debug (GPU box): one soak world on the device, its saved contact rows decoded next to the oracle's LCP.
usage: python tools/dbg/device_world_dbg.py <seed> <world> [big|multi|balls]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import nimblephysics_amd as na
from oracle import OracleWorld
import soak_parity
seed, wd = int(sys.argv[1]), int(sys.argv[2]); mode = sys.argv[3] if len(sys.argv) > 3 else ""
md, s, a, g = soak_parity.make_case(seed, 256, big=mode == "big", multi=mode == "multi", balls=mode == "balls")
if len(sys.argv) > 4:
    import soak_stress
    md, s, a, g = soak_stress.mutator(sys.argv[4])(seed, md, s, a, g)
B = 64
S = np.repeat(s[wd][None], B, 0); A_ = np.repeat(a[wd][None], B, 0)
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = world.to_soa(torch.tensor(S, device="cuda:0")); at = world.to_soa(torch.tensor(A_, device="cuda:0"))
nxt, saved, status = world.step_soa(st, at)
print("device status", hex(int(status[0])), "all equal", bool((status == status[0]).all()))
ow.step(s[wd], a[wd]); L = ow.last_lcp(); m = len(L["b"])
print("oracle status", hex(ow.last_status), "rows", m)
n = md.merge_welds().num_dofs if md.has_welds() else md.num_dofs
MAXR, MAXC, CR = 24, 8, int(os.environ.get("CR_SIZE", "0")) or None
sv = saved.view(torch.float64).cpu().numpy()
import re
src = open(os.path.join(ROOT, "nimblephysics_amd", "csrc", "model_dev.hpp")).read()
CR = int(re.search(r"CR_SIZE = (\d+)", src).group(1)) if CR is None else CR
nc_row = 5 * n; contacts = nc_row + 1; x_row = contacts + MAXC * CR; b_row = x_row + MAXR; cls_row = b_row + MAXR; cfm_row = cls_row + MAXR
total = cfm_row + MAXR + 1 + MAXC
rows = sv[: total * B].reshape(total, B)
nc = int(rows[nc_row, 0]); print("device contacts", nc)
print("device x  ", rows[x_row:x_row + 3 * nc, 0]); print("oracle x  ", L["x"])
print("device b  ", rows[b_row:b_row + 3 * nc, 0]); print("oracle b  ", L["b"])
print("device cls", rows[cls_row:cls_row + 3 * nc, 0]); print("oracle cls", L["row_class"])
print("device cfm", rows[cfm_row:cfm_row + 3 * nc, 0])
dense = 2 * MAXR * MAXR + 2 * n * MAXR
dn = sv[total * B: total * B + dense * B].reshape(B, dense)[0]
Ad = dn[:MAXR * MAXR].reshape(MAXR, MAXR)[:3 * nc, :3 * nc]
Ao_c = L["A"] - (md.fallback_cfm * np.eye(m) if ow.last_status & 0x18 else 0)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lcp_layout import to_device_layout
D = to_device_layout(L, Ao_c, len(ow.last_contacts()))
Ao = D["A"][:3 * nc, :3 * nc]
print("oracle in device layout: rows", D["rows"], "limMask", hex(D["limMask"]), "negMask", hex(D["negMask"]))
print("max |A dev - A oracle|", np.abs(Ad - Ao).max() if D["rows"] == 3 * nc else ("rows differ", 3 * nc, D["rows"]), "scale", np.abs(Ao).max())
print("max |b dev - b oracle|", np.abs(rows[b_row:b_row + 3 * nc, 0] - D["b"][:3 * nc]).max())
np.set_printoptions(linewidth=200, precision=6)
if os.environ.get("PRINT_A"):
    np.set_printoptions(linewidth=200, precision=17)
    print("A device\n", repr(Ad)); print("A oracle\n", repr(Ao)); print("b device", repr(rows[b_row:b_row + 3 * nc, 0])); print("b oracle", repr(D["b"][:3 * nc]))
    np.set_printoptions(linewidth=200, precision=6)
for c in range(nc):
    r0 = contacts + c * CR
    print("contact", c, rows[r0:r0 + CR, 0])

# ---- the host emulation of the same device code on the DEVICE's A and b ----
import ctypes as C
pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
_p = lambda a: a.ctypes.data_as(pd)
_pi = lambda a: a.ctypes.data_as(pi)
shim = C.CDLL(os.path.join(ROOT, "tests", "host_shim", "libcoop_shim.so"))
for name, Ause in (("device A", Ad), ("oracle A", Ao)):
    A24 = np.zeros((24, 24)); A24[:3 * nc, :3 * nc] = Ause; b24 = np.zeros(24); b24[:3 * nc] = rows[b_row:b_row + 3 * nc, 0]
    mu = D["mu"]
    grp = list(range(nc))
    for i in range(nc):
        for j in range(nc):
            if np.abs(A24[3 * i:3 * i + 3, 3 * j:3 * j + 3]).max() > 0:
                gi, gj = grp[i], grp[j]
                grp = [gi if x == gj else x for x in grp]
    for gid in sorted(set(grp)):
      mask = sum(7 << (3 * c) for c in range(nc) if grp[c] == gid)
      print("group", gid, "mask", hex(mask))
      X = np.zeros(24); X0 = np.zeros(24); cls = np.zeros(24, np.int32); E = np.zeros(24)
      ret = shim.shim_coop_stage0_lim(3 * nc, _p(np.ascontiguousarray(A24)), _p(b24), _p(mu), C.c_uint64(mask), C.c_uint64(D["limMask"]), C.c_uint64(D["negMask"]), _p(X), _p(X0), _pi(cls), _p(E))
      Xc = np.zeros(24); Xs = np.zeros(24); cls2 = np.zeros(24, np.int32); cfm = C.c_double(0)
      stg = np.zeros(3, np.int32)
      stt = shim.shim_coop_cascade_lim(3 * nc, _p(np.ascontiguousarray(A24)), _p(b24), _p(mu), _p(X0), C.c_uint64(mask), C.c_uint64(D["limMask"]), C.c_uint64(D["negMask"]), C.c_double(md.fallback_cfm), _p(Xc), C.byref(cfm), _pi(stg))
      print("stage flags", stg)
      print(name, ": sv", np.linalg.svd(Ause, compute_uv=False), "asym", np.abs(Ause - Ause.T).max())
      print(name, ": X0", X0[:3 * nc])
      print(name, ": emulation stage0 ok", ret & 1, "cascade status", hex(stt), "x", Xc[:3 * nc])
print("mu", mu[:nc], [ (bx.mu, bx.body) for bx in md.boxes])
