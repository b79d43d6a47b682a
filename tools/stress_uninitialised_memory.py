"""Stress: alternate Atlas and box-stack steps in one process and look for non-finite outputs (uninitialised-memory hunt)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from util import box_stack_inputs, contact_inputs

def run(md, s, a, g):
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True)
    at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy()
    out.backward(torch.tensor(g, device="cuda:0"))
    return out.detach().cpu().numpy(), st.grad.cpu().numpy(), at.grad.cpu().numpy(), status

ref = {}
for it in range(12):
    # poison the allocator's free blocks
    junk = [torch.full((int(np.random.default_rng(it).integers(1, 40)) * 1000003,), float("nan"), device="cuda:0", dtype=torch.float64) for _ in range(3)]
    del junk
    for tag, (md, s, a) in (("atlas", contact_inputs("atlas20", 1024, 13, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)),
                            ("rim", box_stack_inputs(1024, 16, overhang=True)), ("stack", box_stack_inputs(2048, 15))):
        g = np.random.default_rng(5).normal(0, 1, s.shape)
        o = run(md, s, a, g)
        nf = [int((~np.isfinite(x)).any(1).sum()) for x in o[:3]]
        if tag not in ref: ref[tag] = o
        d = [float(np.abs(np.nan_to_num(x) - np.nan_to_num(y)).max()) for x, y in zip(o[:3], ref[tag][:3])]
        if any(nf) or any(v > 0 for v in d):
            bad = np.where((~np.isfinite(o[0])).any(1))[0]
            print("iter", it, tag, "nonfinite lanes", nf, "diff vs first run", d, "lanes", bad[:5], [hex(x) for x in o[3][bad[:5]]], flush=True)
print("done")
