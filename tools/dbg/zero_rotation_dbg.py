"""Debug (GPU box): exponential-map joints exactly at q = 0 and / or w = 0 (the Taylor branches of expMapRot / expMapJac / logMap and their reverse mode)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from test_ball_joint import ball_model, _ball_offsets
md = ball_model(11, True); n = md.num_dofs; B = 64
offs = [0] + _ball_offsets(md)
for name, zq, zw, tiny in (("q = 0", True, False, 0), ("w = 0", False, True, 0), ("q = w = 0", True, True, 0), ("|q| = 1e-9", True, False, 1e-9), ("|q| = 9.99e-4 / 1.001e-3", True, False, 1e-3)):
    rng = np.random.default_rng(5)
    q = rng.normal(0, 0.5, (B, n)); v = rng.normal(0, 1.0, (B, n))
    for o in offs:
        if zq:
            ax = rng.normal(size=(B, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
            q[:, o:o + 3] = ax * (tiny * (1 + 2e-3 * (np.arange(B)[:, None] % 2 - 0.5)) if tiny else 0.0)
        if zw:
            v[:, o:o + 3] = 0.0
    s = np.concatenate([q, v], 1); a = rng.normal(0, 1, (B, len(md.action_map))); g = rng.normal(0, 1, s.shape)
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at); out.backward(torch.tensor(g, device="cuda:0"))
    ref = OracleWorld(md).step_batch(s, a, g, threads=8)
    sc = lambda x: max(np.abs(x).max(), 1e-30)
    print(f"{name:28s} next {np.abs(out.detach().cpu().numpy() - ref['next']).max() / sc(ref['next']):.1e} grad_state {np.abs(st.grad.cpu().numpy() - ref['grad_state']).max() / sc(ref['grad_state']):.1e} "
          f"grad_action {np.abs(at.grad.cpu().numpy() - ref['grad_action']).max() / sc(ref['grad_action']):.1e} finite {np.isfinite(st.grad.cpu().numpy()).all()}")
