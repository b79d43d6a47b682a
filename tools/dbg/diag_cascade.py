"""Dump the worlds whose device result differs from the oracle's (cascade diagnosis)."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from util import box_stack_inputs, contact_inputs
out = {}
for name, B, seed, kw in [("atlas20", 1024, 13, dict(joint_noise=0.02, vel_noise=0.01, action_noise=0.0)), ("atlas20", 4096, 23, dict(joint_noise=0.02, vel_noise=0.01, action_noise=0.0)), ("box_stack", 8192, 32, {})]:
    if name == "box_stack": md, s, a = box_stack_inputs(B, seed)
    else: md, s, a = contact_inputs(name, B, seed, **kw)
    world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
    g = np.random.default_rng(seed + 1).normal(0, 1, s.shape)
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    o = timestep(world, st, at)
    status = world.last_status.cpu().numpy().astype(np.uint32)
    X = world.lcp_cache.cpu().numpy().T.copy()
    o.backward(torch.tensor(g, device="cuda:0"))
    ref = ow.step_batch(s, a, g, threads=32, want_lcp=True)
    nx = o.detach().cpu().numpy()
    e = np.abs(nx - ref["next"]).max(1) / np.abs(ref["next"]).max()
    eg = np.abs(st.grad.cpu().numpy() - ref["grad_state"]).max(1) / np.abs(ref["grad_state"]).max()
    bad = np.where((e > 1e-9) | (eg > 1e-7))[0]
    print(name, "B", B, "stage0 frac", ((status & 2) != 0).mean(), "n(e>1e-7)", (e > 1e-7).sum(), "n(e>1e-5)", (e > 1e-5).sum(),
          "n(eg>1e-7)", (eg > 1e-7).sum(), "n(eg>1e-5)", (eg > 1e-5).sum(), "status differs", (status != ref["status"]).sum(), flush=True)
    for i in bad[:20]: print("  ", i, e[i], eg[i], hex(status[i]), hex(ref["status"][i]))
    name = f"{name}_{B}_{seed}"
    out.update({f"{name}_idx": bad, f"{name}_s": s[bad], f"{name}_a": a[bad], f"{name}_next": nx[bad], f"{name}_st": status[bad],
                f"{name}_X": X[bad], f"{name}_ost": ref["status"][bad], f"{name}_oX": ref["lcp"][bad], f"{name}_e": e[bad], f"{name}_eg": eg[bad],
                f"{name}_allst": status, f"{name}_allost": ref["status"]})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "diag_cascade.npz"), **out)
