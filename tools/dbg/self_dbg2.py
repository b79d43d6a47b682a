import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from util import folding_arm
import test_gpu_self_collision as t
md = folding_arm(True, "box")
s, a = t._states(512, 1, 1.872, 2.14)
np.set_printoptions(linewidth=200, precision=3)
g = np.random.default_rng(2).normal(0, 1, s.shape)
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
out = timestep(world, st, at); out.backward(torch.tensor(g, device="cuda:0"))
ref = ow.step_batch(s, a, g, threads=8)
e = np.abs(st.grad.cpu().numpy() - ref["grad_state"]).max(1) / np.abs(ref["grad_state"]).max()
bad = np.where(e > 1e-7)[0]
status = world.last_status.cpu().numpy()
print("bad", len(bad))
for wd in bad[:8]:
    ow.reset_lcp_cache(); ow.step(s[wd], a[wd]); c = ow.last_contacts()
    print(wd, f"{e[wd]:.1e}", hex(int(status[wd])), "q", s[wd, :3], "types", c[:, 7], "depth", c[:, 6])
wd = bad[0]
S = np.repeat(s[wd][None], 6, 0); A = np.repeat(a[wd][None], 6, 0); G = np.eye(6)
st = torch.tensor(S, device="cuda:0", requires_grad=True); at = torch.tensor(A, device="cuda:0", requires_grad=True)
out = timestep(world, st, at); out.backward(torch.tensor(G, device="cuda:0"))
ref = ow.step_batch(S, A, G, threads=2)
print(st.grad.cpu().numpy() - ref["grad_state"])
