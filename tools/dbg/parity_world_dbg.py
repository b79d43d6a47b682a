"""One world of a box-stack style batch where the device and the oracle differ: which block, how far from the log-map singularity of the
free joints, what the oracle's own perturbed runs do.  usage: python tools/dbg/parity_world_dbg.py cfg4|twocubes"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from parity import world_errors, entry_scales
from util import box_stack_inputs

which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
if which == "cfg4":
    md, s, a = box_stack_inputs(8192, 32)
    g = np.random.default_rng(33).normal(0, 1, s.shape)
else:
    import test_gpu_contact as T
    raise SystemExit("run the test itself for the two-cube scene")
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
out = timestep(world, st, at)
status = world.last_status.cpu().numpy().astype(np.uint32)
out.backward(torch.tensor(g, device="cuda:0"))
ref = ow.step_batch(s, a, g, threads=8)
dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
errs, scales = world_errors(dev, ref)
worst = np.maximum.reduce(list(errs.values()))
bad = np.where(worst > 1e-7)[0]
print("worlds above 1e-7:", len(bad), "above 1e-5:", int((worst > 1e-5).sum()))
n = md.num_dofs
for wd in bad[:25]:
    rel = {k: np.abs(dev[k][wd] - ref[k][wd]) / scales[k][wd] for k in dev}
    k = max(rel, key=lambda kk: rel[kk].max()); j = int(rel[k].argmax())
    yaws = [abs(s[wd, 1]), abs(s[wd, 7])]
    print(f"world {wd}: status dev {status[wd]:#x} ref {ref['status'][wd]:#x} worst {k}[{j}] rel {rel[k][j]:.2e} dev {dev[k][wd][j]:.6e} ref {ref[k][wd][j]:.6e} scale {scales[k][wd][j]:.3e}; "
          f"pi-|yaw| cubes {np.pi - yaws[0]:.3f} {np.pi - yaws[1]:.3f}; errs " + " ".join(f"{kk}={errs[kk][wd]:.1e}" for kk in errs))
