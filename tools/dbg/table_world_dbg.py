"""Device vs oracle on single worlds of tests/util.table_inputs (the 48-row build): status words and per-output distances to the oracle's
outcomes under ulp perturbations."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from util import table_inputs

md, s, a = table_inputs(512, 51)
g = np.random.default_rng(52).normal(0, 1, s.shape)
world = na.World(md, device="cuda:0")
st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
out = timestep(world, st, at)
status = world.last_status.cpu().numpy().astype(np.uint32)
out.backward(torch.tensor(g, device="cuda:0"))
dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
ow = OracleWorld(md)
ref = ow.step_batch(s, a, g, threads=8)
print("device status histogram", {hex(int(k)): int(v) for k, v in zip(*np.unique(status, return_counts=True))})
print("oracle status histogram", {hex(int(k)): int(v) for k, v in zip(*np.unique(ref["status"], return_counts=True))})
scales = {k: np.abs(ref[k]).max() for k in dev}
err = {k: np.abs(dev[k] - ref[k]).max(1) / scales[k] for k in dev}
worst = np.maximum.reduce(list(err.values()))
bad = np.where(worst > 1e-7)[0]
print("worlds above 1e-7:", len(bad), bad[:20])
rng = np.random.default_rng(1)
for wd in bad[:12]:
    sp = s[wd][None] * (1 + rng.integers(-16, 17, (64, s.shape[1])) * 2.2e-16)
    rr = ow.step_batch(sp, np.repeat(a[wd:wd + 1], 64, 0), np.repeat(g[wd:wd + 1], 64, 0), threads=8)
    sts = {hex(int(k)): int(v) for k, v in zip(*np.unique(rr["status"], return_counts=True))}
    d = {k: (np.abs(rr[k] - dev[k][wd][None]).max(1) / scales[k]) for k in dev}
    print(f"world {wd}: device {hex(int(status[wd]))} oracle {hex(int(ref['status'][wd]))} perturbed oracle {sts}; err vs unperturbed "
          + " ".join(f"{k}={err[k][wd]:.2e}" for k in dev) + "; min dist to a perturbed run " + " ".join(f"{k}={d[k].min():.2e}" for k in dev))
    for stv in np.unique(rr["status"]):
        sel = rr["status"] == stv
        print("    outcome", hex(int(stv)), "min dist:", " ".join(f"{k}={d[k][sel].min():.2e}" for k in dev))
