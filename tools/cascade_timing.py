"""Debug: per-stage cycle stamps of the cooperative cascade.  Needs the library built with the stamps compiled in:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DNBL_CASCADE_TIMING nimblephysics_amd/csrc/nimble_amd.hip \\
        -o tools/dbg/libnimble_amd_timing.so
usage (GPU box): python tools/cascade_timing.py <joint noise>"""
import os, sys, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nimblephysics_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "dbg", "libnimble_amd_timing.so")
import nimblephysics_amd as na
from util import contact_inputs
jn = float(sys.argv[1]) if len(sys.argv) > 1 else 0.005
md, s, a = contact_inputs("atlas20", 4096, 1000, joint_noise=jn, vel_noise=jn / 2, action_noise=0.1)
world = na.World(md, device="cuda:0")
B = 4096
st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
import ctypes
L = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
L.nbl_debug_dantzig_stats(buf, 1)
nxt, saved, status = world.step_soa(st, at)
torch.cuda.synchronize()
L.nbl_debug_dantzig_stats(buf, 0)
dz = list(buf)
names_dz = ["setup", "w[i]", "solve1", "dw", "step selection", "apply + transfer", "(loop tail)"]
calls, idx, piv, rem = dz[8], dz[9], dz[10], dz[11]
print(f"Dantzig: {calls} solves, {idx / max(calls, 1):.1f} driving rows, {piv / max(calls, 1):.1f} pivot iterations, {rem / max(calls, 1):.1f} C->N removals per solve")
for k, nm in enumerate(names_dz):
    print(f"  {nm:18s} {dz[k] / max(calls, 1):10.0f} cycles per solve")
for k, nm in ((12, "stage 1: load problem"), (13, "stage 1: reduce"), (14, "stage 1: Dantzig (all of it)"), (15, "stage 1: map out + validity")):
    print(f"  {nm:30s} {dz[k] / max(calls, 1):10.0f} cycles per solve")
stat = status.cpu().numpy()
ws = world._workspace(B).view(torch.float64).cpu().numpy()
nb = 15
lws = ws[nb * 288 * B:]
LW_JB = 144
base = LW_JB + 3 * 24 + 3 * 8 + 1          # LW_STAGE_CYCLES (model_dev.hpp, coop_kernels.hip)
rows = lws[: (lws.size // B) * B].reshape(-1, B)[base:base + 4]
failed = np.where((stat & 0x2) == 0)[0]
print("failed worlds", len(failed))
names = ["stage 1 wave (reduce + Dantzig + validity)", "stage 2 wave (CFM: reduce + PGS + validity)", "stage 3 wave (no friction: PGS)", "final kernel (select + standardise + outputs)"]
for k, nme in enumerate(names):
    d = rows[k, failed]
    print(f"{nme:48s} mean {d.mean():9.0f}  p50 {np.percentile(d, 50):9.0f}  p90 {np.percentile(d, 90):9.0f}  p99 {np.percentile(d, 99):9.0f}  max {d.max():9.0f} cycles")
crit = np.maximum.reduce([rows[0, failed], rows[1, failed], rows[2, failed]])
print(f"{'longest stage wave of a world':48s} mean {crit.mean():9.0f}  p90 {np.percentile(crit, 90):9.0f}  max {crit.max():9.0f}")
import collections
print(collections.Counter(hex(x) for x in stat[failed]))
