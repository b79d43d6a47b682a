"""developer aid: the general contact route step by step (forward sync, backward sync) on the tower / three-group scenes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import nimblephysics_amd as na
from oracle import OracleWorld
from util import cube_tower_inputs
from parity import world_errors

which = sys.argv[1] if len(sys.argv) > 1 else "tower10"
if which.startswith("tower"):
    nc = int(which[5:])
    md, s, a = cube_tower_inputs(8, 7 + nc, nc, max_contacts=4 * nc + 8)
else:
    import test_gpu_general as tg          # groups<towers><table 0/1>, e.g. groups21
    md, s, a = tg.three_groups_scene(8, int(which[6]), which[7] == "1")
world = na.World(md, device="cuda:0")
print("max contacts of the build:", world._L.nbl_model_max_contacts(world._h), "n", world.n, flush=True)
st = torch.tensor(s, device="cuda:0"); at = torch.tensor(a, device="cuda:0")
nxt, sv, status = world.step_soa(world.to_soa(st), world.to_soa(at), want_saved=True)
torch.cuda.synchronize()
print("forward done; status", [hex(int(x)) for x in status.cpu().numpy()[:8]], flush=True)
g = np.random.default_rng(2).normal(0, 1, s.shape)
ow = OracleWorld(md)
ref = ow.step_batch(s, a, g, threads=4)
nx = world.from_soa(nxt).cpu().numpy()
print("oracle status", [hex(int(x)) for x in ref["status"][:8]])
print("next err per world", (np.abs(nx - ref["next"]).max(1) / np.abs(ref["next"]).max(1)))
gs, ga = world.backward_soa(sv, world.to_soa(torch.tensor(g, device="cuda:0")))
torch.cuda.synchronize()
print("backward done", flush=True)
dev = {"next": nx, "grad_state": world.from_soa(gs).cpu().numpy(), "grad_action": world.from_soa(ga).cpu().numpy()}
e, _ = world_errors(dev, ref)
for k in e:
    print(k, e[k])
