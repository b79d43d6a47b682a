import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from test_gpu_random_trees import random_tree
for shape in ("chain", "star", "random"):
    rng = np.random.default_rng(1)
    md = random_tree(rng, 12, shape, True, colliders=5, spheres=True, balls=1.0)
    for b in md.bodies[1:]:
        b.joint_type = "ball"; b.damping = (); b.spring = (); b.rest = ()
    md = na.ModelDescription("limit", md.bodies, md.boxes, gravity=md.gravity, dt=md.dt, max_contacts=8)
    n = md.num_dofs; B = 128
    q = rng.normal(0, 0.3, (B, n)); q[:, 3] = rng.normal(0, 0.3, B); q[:, 5] = rng.normal(0, 0.3, B); q[:, 4] = rng.uniform(0.02, 0.5, B)
    v = rng.normal(0, 0.5, (B, n)); s = np.concatenate([q, v], 1); a = rng.normal(0, 0.5, (B, len(md.action_map))); g = rng.normal(0, 1, s.shape)
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at); status = world.last_status.cpu().numpy().astype(np.uint32); out.backward(torch.tensor(g, device="cuda:0"))
    ref = OracleWorld(md).step_batch(s, a, g, threads=8)
    dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
    err = np.maximum.reduce([np.abs(dev[k] - ref[k]).max(1) / max(np.abs(ref[k]).max(), 1e-30) for k in dev])
    err[((status | ref["status"]) & 0x80) != 0] = 0
    print(shape, "dofs", n, "device bodies", 1 + 3 * 11, "in contact", (status & 1).mean(), "overflow", ((status & 0x80) != 0).mean(), "max err", err.max(), "> 1e-5:", int((err > 1e-5).sum()))
