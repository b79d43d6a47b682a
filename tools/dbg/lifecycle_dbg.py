import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, gc
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from util import contact_inputs
from test_ball_joint import ball_model
torch.cuda.init()
free0, total = torch.cuda.mem_get_info()
md, s, a = contact_inputs("atlas20", 256, 1)
mb = ball_model(1, True, ground=True)
ref = None
for it in range(120):
    m = md if it % 2 == 0 else mb
    w = na.World(m, device="cuda:0")
    n = m.num_dofs
    x = torch.tensor(s if it % 2 == 0 else np.random.default_rng(it).normal(0, 0.3, (64, 2 * n)), device="cuda:0", requires_grad=True)
    u = torch.tensor(a if it % 2 == 0 else np.zeros((64, len(m.action_map))), device="cuda:0")
    y = timestep(w, x, u); y.sum().backward()
    if it == 0: ref = y.detach().clone()
    if it % 2 == 0: assert torch.equal(ref, y.detach())
    w2 = w.clone(); del w, w2, x, y
    if it % 20 == 19:
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
        print(it + 1, "worlds created / destroyed; device memory in use by others than torch's cache (MB):", (total - torch.cuda.mem_get_info()[0]) // 2**20)
