"""Debug (GPU box): device vs oracle as the whole scene moves away from the world origin (world-frame spatial algebra loses digits with distance)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from util import contact_inputs, cfg_inputs
for name, contact in (("atlas20", False), ("atlas20", True)):
    for shift in (0.0, 10.0, 100.0, 1000.0, 1e4):
        if contact:
            md, s, a = contact_inputs("atlas20", 256, 5)
            for bx in md.boxes:                       # move the ground along
                if bx.body < 0 or md.bodies[bx.body].joint_type == "weld":
                    pass
        else:
            md, s, a = cfg_inputs("atlas20", 256, 5)
        s = s.copy(); s[:, 3] += shift; s[:, 5] += shift
        g = np.random.default_rng(1).normal(0, 1, s.shape)
        world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
        st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
        out = timestep(world, st, at); out.backward(torch.tensor(g, device="cuda:0"))
        ref = ow.step_batch(s, a, g, threads=8)
        n = s.shape[1] // 2
        # compare velocities and gradients (positions carry the shift itself)
        e_v = np.abs(out.detach().cpu().numpy()[:, n:] - ref["next"][:, n:]).max() / np.abs(ref["next"][:, n:]).max()
        e_g = np.abs(st.grad.cpu().numpy() - ref["grad_state"]).max() / np.abs(ref["grad_state"]).max()
        stat = world.last_status.cpu().numpy()
        print(f"{name} contact={contact} shift {shift:8.0f}: next-velocity err {e_v:.2e} grad_state err {e_g:.2e} in contact {(stat & 1).mean():.2f} status equal {(stat.astype(np.uint32) == ref['status']).mean():.3f}")
