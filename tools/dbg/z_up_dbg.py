"""Debug (GPU box): a z-up scene (gravity -z, ground normal +-z): the tangent basis of ContactConstraint::getTangentBasisMatrixODE takes its
fallback branch (normal parallel to the first candidate axis z)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from test_gpu_random_trees import random_tree
tot = bad = 0
for seed in range(40):
    rng = np.random.default_rng(7000 + seed)
    md = random_tree(rng, int(rng.integers(1, 6)), "random", True, colliders=int(rng.integers(1, 4)), spheres=bool(rng.random() < 0.5), balls=0.3)
    md.gravity = (0.0, 0.0, -9.81)
    md.boxes[0] = na.BoxSpec(-1, na.make_transform((0, 0, -0.005)), (20.0, 20.0, 0.01), 1.0)
    B = 128; n = md.num_dofs
    q = rng.normal(0, 0.25, (B, n)); q[:, 3] = rng.normal(0, 0.3, B); q[:, 4] = rng.normal(0, 0.3, B); q[:, 5] = rng.uniform(0.02, 0.4, B)
    v = rng.normal(0, 0.5, (B, n))
    s = np.concatenate([q, v], 1); a = rng.normal(0, 0.5, (B, len(md.action_map))); g = rng.normal(0, 1, s.shape)
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at); status = world.last_status.cpu().numpy().astype(np.uint32); out.backward(torch.tensor(g, device="cuda:0"))
    ref = OracleWorld(md).step_batch(s, a, g, threads=8)
    dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
    err = np.maximum.reduce([np.abs(dev[k] - ref[k]).max(1) / max(np.abs(ref[k]).max(), 1e-30) for k in dev])
    err[((status | ref["status"]) & 0x80) != 0] = 0
    tot += int((status & 1).sum()); bad += int((err > 1e-5).sum())
    if (err > 1e-5).any(): print("seed", seed, "worlds > 1e-5:", int((err > 1e-5).sum()), "max", err.max(), "contact", (status & 1).mean())
print("worlds in contact", tot, "above 1e-5:", bad)
