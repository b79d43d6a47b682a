import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from oracle import OracleWorld
from util import limited_arm
import test_gpu_joint_limits as t
md = limited_arm(ground=True)
s, a = t._states(md, 1024, 5, at_limit=0.35)
rng = np.random.default_rng(6)
s[:, 0] = rng.uniform(-0.025, 0.008, len(s))
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
n1, saved, status = world.step_soa(st, at)
cache = world.lcp_cache.cpu().numpy().T.copy()
ref = ow.step_batch(s, a, None, threads=8, want_lcp=True)
np.set_printoptions(linewidth=200, precision=6)
for wd in (116, 25, 60):
    print(wd, hex(int(status[wd])), "dev rows", cache[wd, 24], "ref rows", ref["lcp_len"][wd])
    print(" dev", cache[wd, :24])
    print(" ref", ref["lcp"][wd][:ref["lcp_len"][wd]])
