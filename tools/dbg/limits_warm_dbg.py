import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from oracle import OracleWorld
from util import limited_arm
import test_gpu_joint_limits as t
md = limited_arm()
s, a = t._states(md, 256, 8, at_limit=0.6)
s[:, md.num_dofs:] *= 0.05
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
n1, _, status1 = world.step_soa(st, at)
cache = world.lcp_cache.cpu().numpy().T.copy()
rows = cache[:, 24].astype(int)
n2, _, status2 = world.step_soa(n1, at)
cache2 = world.lcp_cache.cpu().numpy().T.copy()
s1 = world.from_soa(n1).cpu().numpy()
ref = ow.step_batch(s1, a, None, threads=8, lcp_in=np.concatenate([cache[:, 0:24:3], np.zeros((len(s), 16))], 1), lcp_len_in=rows // 3, want_lcp=True)
err = np.abs(world.from_soa(n2).cpu().numpy() - ref["next"]).max(1)
bad = np.where(err > 1e-9)[0]
st2 = status2.cpu().numpy().astype(np.uint32)
np.set_printoptions(linewidth=200, precision=5)
print("bad", len(bad), err.max())
for wd in bad[:6]:
    print(wd, f"{err[wd]:.2e}", hex(int(status1[wd])), hex(st2[wd]), hex(ref["status"][wd]), "rows1", rows[wd], "rows2", cache2[wd, 24], "ref len", ref["lcp_len"][wd])
    print("  cache1", cache[wd, :rows[wd]:3]); print("  dev x2 ", cache2[wd, :int(cache2[wd,24]):3]); print("  ref x2 ", ref["lcp"][wd][:ref["lcp_len"][wd]])
    print("  q1", s1[wd, :5])
