"""Debug: per-stage cycle stamps of the cooperative cascade.  Needs the library built with the stamps compiled in:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DNBL_CASCADE_TIMING nimblephysics_amd/csrc/nimble_amd.hip \\
        -o tools/dbg/libnimble_amd_timing.so
usage (GPU box): python tools/cascade_timing.py <joint noise> [worlds]"""
import os, sys, shutil, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nimblephysics_amd._lib as _lib
_lib.LIB_PATH = os.environ.get("NBL_TIMING_LIB", os.path.join(ROOT, "tools", "dbg", "libnimble_amd_timing.so"))
import nimblephysics_amd as na
from util import contact_inputs
jn = float(sys.argv[1]) if len(sys.argv) > 1 else 0.005
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096      # 1024 = one stream slice of the bench: ~1 stage wavefront per SIMD, like in the timed run
md, s, a = contact_inputs("atlas20", B, 1000, joint_noise=jn, vel_noise=jn / 2, action_noise=0.1)
world = na.World(md, device="cuda:0")
if len(sys.argv) > 2:
    world.set_slices(1)
st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
import ctypes
L = _lib.lib()
buf = (ctypes.c_ulonglong * 24)()
L.nbl_debug_dantzig_stats(buf, 1)
nxt, saved, status = world.step_soa(st, at)
torch.cuda.synchronize()
L.nbl_debug_dantzig_stats(buf, 0)
dz = list(buf)
names_dz = ["setup", "w[i]", "solve1", "dw", "step selection", "apply + row enters C / N", "(loop tail)", "apply + row leaves C"]
calls, idx, piv, rem = dz[8], dz[9], dz[10], dz[11]
print(f"Dantzig: {calls} solves, {idx / max(calls, 1):.1f} driving rows, {piv / max(calls, 1):.1f} pivot iterations, {rem / max(calls, 1):.1f} C->N removals per solve")
for k, nm in enumerate(names_dz):
    print(f"  {nm:26s} {dz[k] / max(calls, 1):10.0f} cycles per solve")
if hasattr(L, "nbl_debug_dantzig_stats_slow"):
    sb = (ctypes.c_ulonglong * 16)()
    L.nbl_debug_dantzig_stats_slow(sb)
    sl = list(sb)
    if sl[8]:
        c2 = sl[8]
        print(f"the SLOW solves (the tail a launch waits for): {c2} solves, {sl[9] / c2:.1f} driving rows, {sl[10] / c2:.1f} pivot iterations, {sl[11] / c2:.1f} C->N removals per solve")
        for k, nm in enumerate(names_dz):
            print(f"  {nm:26s} {sl[k] / c2:10.0f} cycles per solve")
for k, nm in ((12, "stage 1: load problem"), (13, "stage 1: reduce"), (14, "stage 1: Dantzig (all of it)"), (15, "stage 1: map out + validity")):
    print(f"  {nm:30s} {dz[k] / max(calls, 1):10.0f} cycles per solve")
pv = dz[16:24]
if pv[0]:
    print(f"Householder route: {pv[0]} factorisations; cycles each: pivoted QR {pv[1] / pv[0]:.0f}, R1^-1 [R2 | G1] {pv[2] / pv[0]:.0f}, W W^T {pv[3] / pv[0]:.0f}, "
          f"Cholesky of I + W W^T {pv[4] / pv[0]:.0f}, two substitutions {pv[5] / pv[0]:.0f}, [z; W^T z] {pv[6] / pv[0]:.0f}")
stat = status.cpu().numpy()
ws = world._workspace(B).view(torch.float64).cpu().numpy()
nb = 15
lws = ws[nb * 288 * B:]
LW_JB = 144
base = LW_JB + 3 * 24 + 3 * 8 + 1          # LW_STAGE_CYCLES (model_dev.hpp, coop_kernels.hip)
rows = lws[: (lws.size // B) * B].reshape(-1, B)[base:base + 9]
failed = np.where((stat & 0x2) == 0)[0]
print("failed worlds", len(failed))
names = ["stage 1 wave (reduce + Dantzig + validity)", "stage 2 wave (CFM: reduce + PGS + validity)", "stage 3 wave (no friction: PGS)", "final kernel (select + standardise + outputs)"]
for k, nme in enumerate(names):
    d = rows[k, failed]
    print(f"{nme:48s} mean {d.mean():9.0f}  p50 {np.percentile(d, 50):9.0f}  p90 {np.percentile(d, 90):9.0f}  p99 {np.percentile(d, 99):9.0f}  max {d.max():9.0f} cycles")
d = rows[4]
ok = np.where((stat & 0x2) != 0)[0]
for nme, sel in (("stage-0 kernel, all worlds", slice(None)), ("stage-0 kernel, worlds it resolves", ok), ("stage-0 kernel, worlds it hands on", failed)):
    x = d[sel]
    print(f"{nme:48s} mean {x.mean():9.0f}  p50 {np.percentile(x, 50):9.0f}  p90 {np.percentile(x, 90):9.0f}  p99 {np.percentile(x, 99):9.0f}  max {x.max():9.0f} cycles")
con = np.where((stat & 0x1) != 0)[0]
g, lp, packed, fast = rows[5][con], rows[6][con], rows[7][con].astype(int), rows[8][con]
nu, ncl, rk = packed % 100, (packed // 100) % 100, packed // 10000
hh = nu > 0
print("Householder-route worlds: clamping rows (nc) histogram", dict(collections.Counter(ncl[hh].tolist())), "rank histogram", dict(collections.Counter(rk[hh].tolist())))
okc = ((stat & 0x2) != 0)[con]
print(f"worlds with contacts {len(con)}: guess (load + factorisation + apply) mean {g.mean():.0f} p90 {np.percentile(g, 90):.0f} max {g.max():.0f}")
for nme, sel in (("final classification = the guess rows (no second factorisation)", fast == 1), ("other classification, no friction row on its bound (Cholesky route)", (fast != 1) & (nu == 0)),
                 ("friction rows on their bound (Householder route), Q of full rank", (nu > 0) & (fast == 0)),
                 ("friction rows on their bound (Householder route), Q rank deficient", (nu > 0) & (fast == 2))):
    for tag, sel2 in (("resolved", sel & okc), ("handed on", sel & ~okc)):
        if sel2.sum():
            x = lp[sel2]
            print(f"  {nme:70s} {tag:9s} {sel2.sum():5d} worlds: loop mean {x.mean():8.0f} p50 {np.percentile(x, 50):8.0f} p90 {np.percentile(x, 90):8.0f} max {x.max():8.0f}")
crit = np.maximum.reduce([rows[0, failed], rows[1, failed], rows[2, failed]])
print(f"{'longest stage wave of a world':48s} mean {crit.mean():9.0f}  p90 {np.percentile(crit, 90):9.0f}  max {crit.max():9.0f}")
w1 = rows[1, failed] + rows[2, failed]            # the PGS wavefront runs stage 2, then (when stage 2 is not valid) stage 3
both = np.maximum(rows[0, failed], w1)
print(f"{'PGS wavefront (stage 2 + stage 3)':48s} mean {w1.mean():9.0f}  p90 {np.percentile(w1, 90):9.0f}  p99 {np.percentile(w1, 99):9.0f}  max {w1.max():9.0f}")
print(f"{'the slower of the two wavefronts of a world':48s} mean {both.mean():9.0f}  p90 {np.percentile(both, 90):9.0f}  p99 {np.percentile(both, 99):9.0f}  max {both.max():9.0f};  "
      f"worlds whose PGS wavefront is the slower one: {float((w1 > rows[0, failed]).mean()):.2f}; the ten slowest worlds (Dantzig, PGS): "
      f"{[(int(rows[0, failed][i]), int(w1[i])) for i in np.argsort(-both)[:10]]}")
import collections
print(collections.Counter(hex(x) for x in stat[failed]))
fin = rows[3, failed]
order = np.argsort(-fin)[:12]
print("slowest worlds of the final kernel (cycles, status, share of the failed worlds above 70 k / 100 k cycles):", [(int(fin[i]), hex(stat[failed[i]])) for i in order],
      float((fin > 70e3).mean()), float((fin > 100e3).mean()))
for bits in sorted(set(stat[failed].tolist())):
    x = fin[stat[failed] == bits]
    print(f"   status {bits:#x}: {len(x):5d} worlds, final kernel mean {x.mean():8.0f} p90 {np.percentile(x, 90):8.0f} max {x.max():8.0f}")
