import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import nimblephysics_amd as na
import soak_parity, soak_stress
from oracle import OracleWorld
seed, wd = 46203, 141
md, s, a, g = soak_parity.make_case(seed, 256, balls=True)
md, s, a, g = soak_stress.mutator("selfcol")(seed, md, s, a, g)
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
S = np.repeat(s[wd][None], 4, 0); A = np.repeat(a[wd][None], 4, 0)
st = world.to_soa(torch.tensor(S, device="cuda:0")); at = world.to_soa(torch.tensor(A, device="cuda:0"))
nxt, saved, status = world.step_soa(st, at)
cache = world.lcp_cache.cpu().numpy().T
ref = ow.step_batch(S, A, None, threads=1, want_lcp=True)
np.set_printoptions(linewidth=220, precision=5)
print("dev", hex(int(status[0])), "ref", hex(ref["status"][0]))
print("dev x", cache[0, :13]); print("ref x", ref["lcp"][0][:12])
print("dnext", world.from_soa(nxt).cpu().numpy()[0] - ref["next"][0])
