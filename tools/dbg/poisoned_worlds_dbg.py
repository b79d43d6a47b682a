import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from util import contact_inputs
md, s, a = contact_inputs("atlas20", 64, 1, joint_noise=0.02)
s = s.copy(); a = a.copy()
s[3, 7] = np.nan; s[5, 25] = np.inf; a[9, 2] = np.nan; s[11, 4] = 1e300      # NaN position, Inf velocity, NaN torque, a body 1e300 m up
world = na.World(md, device="cuda:0")
st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
out = timestep(world, st, at); out.sum().backward(); torch.cuda.synchronize()
status = world.last_status.cpu().numpy().astype(np.uint32)
o = out.detach().cpu().numpy(); gs = st.grad.cpu().numpy()
bad = [3, 5, 9, 11]; good = [i for i in range(64) if i not in bad]
print("statuses of the poisoned worlds:", [hex(int(status[i])) for i in bad])
print("healthy worlds finite:", bool(np.isfinite(o[good]).all() and np.isfinite(gs[good]).all()), "poisoned worlds' outputs finite:", [bool(np.isfinite(o[i]).all()) for i in bad])
md2, s2, a2 = contact_inputs("atlas20", 64, 1, joint_noise=0.02)
w2 = na.World(md2, device="cuda:0"); o2 = timestep(w2, torch.tensor(s2, device="cuda:0"), torch.tensor(a2, device="cuda:0")).cpu().numpy()
print("healthy worlds identical to a run without the poisoned ones:", bool(np.array_equal(o[good], o2[good])))
