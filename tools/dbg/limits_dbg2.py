import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from util import limited_arm
import test_gpu_joint_limits as t
from parity import world_errors, KEYS
for enforce in (False, True):
    md = limited_arm(ground=True, enforce=enforce)
    s, a = t._states(limited_arm(ground=True), 1024, 5, at_limit=0.35)
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
    errs, scales = world_errors(dev, ref)
    worst = np.maximum.reduce([errs[k] for k in KEYS])
    bad = np.where(worst > 1e-7)[0]
    print("enforce", enforce, "bad", len(bad), "max", worst.max(), {k: float(errs[k].max()) for k in KEYS})
    print([(int(b), f"{worst[b]:.1e}", hex(status[b])) for b in bad[:24]])
    if enforce and len(bad):
        wd = 116
        print("world", wd, hex(status[wd]), s[wd])
        for k in KEYS: print(k, dev[k][wd] - ref[k][wd])
