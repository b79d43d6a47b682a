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
nxt, saved, status = world.step_soa(st, at)
torch.cuda.synchronize()
stat = status.cpu().numpy()
ws = world._workspace(B).view(torch.float64).cpu().numpy()
nb = 15
lws = ws[nb * 288 * B:]
LW_JB = 144
rows = lws[: (lws.size // B) * B].reshape(-1, B)[LW_JB:LW_JB + 8]
failed = np.where((stat & 0x2) == 0)[0]
print("failed worlds", len(failed))
names = ["reduce", "dantzig", "valid1", "stage2", "stage3", "standardise", "outputs"]
d = np.diff(rows[:, failed], axis=0)
for k, nme in enumerate(names):
    print(f"{nme:12s} mean {d[k].mean():10.0f} max {d[k].max():10.0f} cycles")
print("total mean", rows[7, failed].mean(), "max", rows[7, failed].max())
import collections
print(collections.Counter(hex(x) for x in stat[failed]))
