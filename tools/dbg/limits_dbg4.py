import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from util import limited_arm
import test_gpu_joint_limits as t
md = limited_arm(ground=True)
s, a = t._states(md, 1024, 5, at_limit=0.35)
rng = np.random.default_rng(6)
s[:, 0] = rng.uniform(-0.025, 0.008, len(s))
wd = int(sys.argv[1]) if len(sys.argv) > 1 else 116
n2 = s.shape[1]
S = np.repeat(s[wd][None], n2, 0); A = np.repeat(a[wd][None], n2, 0); G = np.eye(n2)
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = torch.tensor(S, device="cuda:0", requires_grad=True); at = torch.tensor(A, device="cuda:0", requires_grad=True)
out = timestep(world, st, at)
out.backward(torch.tensor(G, device="cuda:0"))
ref = ow.step_batch(S, A, G, threads=8)
np.set_printoptions(linewidth=220, precision=3)
D = st.grad.cpu().numpy() - ref["grad_state"]
print("rows: cotangent on next-state component k; columns: d/d state.  device - oracle:")
print(D)
print("oracle J^T:"); print(ref["grad_state"])
