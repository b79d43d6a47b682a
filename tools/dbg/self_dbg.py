import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from util import folding_arm
md = folding_arm(True, "box")
np.set_printoptions(linewidth=200, precision=3)
for q2 in (1.8825, 1.9725, 2.0325, 2.1):
    s0 = np.array([0.3, 2.1, q2, 0.1, -0.2, 0.15]); a0 = np.array([0.05, -0.02, 0.03])
    n2 = 6
    S = np.repeat(s0[None], n2, 0); A = np.repeat(a0[None], n2, 0); G = np.eye(n2)
    world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
    st = torch.tensor(S, device="cuda:0", requires_grad=True); at = torch.tensor(A, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at); out.backward(torch.tensor(G, device="cuda:0"))
    ref = ow.step_batch(S, A, G, threads=2)
    ow.reset_lcp_cache(); ow.step(s0, a0); c = ow.last_contacts()
    print("q2", q2, "status", hex(int(world.last_status[0])), hex(ref["status"][0]), "types", c[:, 7], "next err", np.abs(out.detach().cpu().numpy() - ref["next"]).max())
    print(st.grad.cpu().numpy() - ref["grad_state"])
