import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from util import box_stack_inputs
B = 512
for mus in [(1.0, 0.0005, 1.0), (0.0, 0.0, 0.0)]:
    md, s, a = box_stack_inputs(B, 77)
    for bx, mu in zip(md.boxes, mus): bx.mu = mu
    world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
    g = np.random.default_rng(78).normal(0, 1, s.shape)
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at); status = world.last_status.cpu().numpy().astype(np.uint32)
    out.backward(torch.tensor(g, device="cuda:0"))
    ref = ow.step_batch(s, a, g, threads=8)
    dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
    sc = {k: np.abs(ref[k]).max() for k in dev}
    errs = {k: np.abs(dev[k] - ref[k]).max(1) / max(sc[k], 1e-30) for k in dev}
    bad = np.where(np.maximum.reduce(list(errs.values())) > 1e-5)[0]
    print(mus, "bad worlds", len(bad), {k: float(v.max()) for k, v in errs.items()}, "status hist", {hex(k): int(v) for k, v in zip(*np.unique(status, return_counts=True))})
    rng = np.random.default_rng(12345)
    for wd in bad[:12]:
        sp = s[wd][None, :] * (1.0 + rng.choice([-1.0, 0.0, 1.0], (64, s.shape[1])) * 2.220446049250313e-16)
        r = ow.step_batch(sp, np.repeat(a[wd][None], 64, 0), np.repeat(g[wd][None], 64, 0), threads=8)
        d = {k: np.abs(r[k] - dev[k][wd][None]).max(1) / sc[k] for k in dev}
        spread = {k: float(np.abs(r[k] - ref[k][wd][None]).max() / sc[k]) for k in dev}
        j = np.argmin(np.maximum.reduce(list(d.values())))
        print("  world", wd, hex(status[wd]), hex(ref["status"][wd]), "errs", {k: float(errs[k][wd]) for k in errs}, "nearest", {k: float(d[k][j]) for k in d}, "spread", spread, "ostat", set(hex(x) for x in r["status"]))
