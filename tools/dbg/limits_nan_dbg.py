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
g = np.random.default_rng(7).normal(0, 1, s.shape)
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
out = timestep(world, st, at)
status = world.last_status.cpu().numpy().astype(np.uint32)
out.backward(torch.tensor(g, device="cuda:0"))
ref = ow.step_batch(s, a, g, threads=8)
dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
for k in dev:
    bad = ~np.isfinite(dev[k]).all(1)
    print(k, bad.sum(), np.where(bad)[0][:8], [hex(x) for x in status[bad][:8]], [hex(x) for x in ref["status"][bad][:8]])
    err = np.abs(np.nan_to_num(dev[k]) - ref[k]).max(1)
    print("  max err finite", err[~bad].max())
i = np.where(~np.isfinite(dev["grad_state"]).all(1))[0]
if len(i):
    w0 = i[0]
    print(s[w0]); print(dev["grad_state"][w0]); print(ref["grad_state"][w0]); print(dev["next"][w0]-ref["next"][w0])
from parity import world_errors, KEYS
errs, scales = world_errors(dev, ref)
worst = np.maximum.reduce([errs[k] for k in KEYS])
bad = np.where(worst > 1e-7)[0]
print("bad worlds", len(bad), [(int(b), f"{worst[b]:.1e}", hex(status[b])) for b in bad[:20]])
EPS = 2.220446049250313e-16
rng = np.random.default_rng(1)
for wd in bad[:12]:
    for npert in (64, 512):
        sp = s[wd][None] * (1.0 + rng.integers(-1, 2, (npert, s.shape[1])) * EPS)
        r = ow.step_batch(sp, np.repeat(a[wd][None], npert, 0), np.repeat(g[wd][None], npert, 0), threads=8)
        dist = np.maximum.reduce([np.abs(r[k] - dev[k][wd][None]).max(1) / scales[k] for k in KEYS])
        spread = max(np.abs(r[k] - ref[k][wd][None]).max() / scales[k] for k in KEYS)
        print(int(wd), npert, f"dist {dist.min():.2e} spread {spread:.2e}", end=" | ")
    print()
