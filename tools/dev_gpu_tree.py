"""Debug: coop tree kernels (NBL_COOP_TREE=1) vs one-world-per-lane tree kernels on the box-stack rim case."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from util import box_stack_inputs

def run(coop_tree, md, s, a, g):
    os.environ["NBL_COOP_TREE"] = "1" if coop_tree else "0"
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True)
    at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy()
    out.backward(torch.tensor(g, device="cuda:0"))
    return out.detach().cpu().numpy(), st.grad.cpu().numpy(), at.grad.cpu().numpy(), status

for overhang in (True, False):
    md, s, a = box_stack_inputs(1024, 41, overhang=overhang)
    g = np.random.default_rng(5).normal(0, 1, s.shape)
    for rep in range(3):
        o1 = run(True, md, s, a, g)
        o0 = run(False, md, s, a, g)
        bad = np.where(~np.isfinite(o1[0]).all(1))[0]
        e = np.abs(np.nan_to_num(o1[0]) - o0[0]).max(1)
        print("overhang", overhang, "rep", rep, "nan lanes", bad[:10], "status", [hex(x) for x in o1[3][bad[:5]]], [hex(x) for x in o0[3][bad[:5]]],
              "max diff other lanes", e[np.isfinite(o1[0]).all(1)].max(), "grad diff", np.abs(np.nan_to_num(o1[1]) - o0[1]).max())
        if len(bad):
            i = bad[0]
            print("  lane", i, "state", np.round(s[i], 4), "\n  next coop", o1[0][i], "\n  next lane", o0[0][i])
