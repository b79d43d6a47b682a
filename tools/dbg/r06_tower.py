import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import nimblephysics_amd._lib as _lib
if os.environ.get("NBL_DBG_LIB"): _lib.LIB_PATH = os.environ["NBL_DBG_LIB"]
import nimblephysics_amd as na
from util import cube_tower_inputs
n_cubes = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
md, s, a = cube_tower_inputs(B, 7 + n_cubes, n_cubes, max_contacts=4 * n_cubes + 8)
world = na.World(md, device="cuda:0")
st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
nxt, saved, status = world.step_soa(st, at); torch.cuda.synchronize()
print("ok", np.unique(status.cpu().numpy(), return_counts=True))
