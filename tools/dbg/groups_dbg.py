import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nimblephysics_amd as na
from oracle import OracleWorld
from util import contact_inputs
md, s, a = contact_inputs("atlas20", 256, 5, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
nxt, saved, status = world.step_soa(st, at)
status = status.cpu().numpy().astype(np.uint32)
ref = ow.step_batch(s, a, None, threads=8)
import collections
print("device", collections.Counter(hex(x) for x in status).most_common(6))
print("oracle", collections.Counter(hex(x) for x in ref["status"]).most_common(6))
print("next err", np.abs(world.from_soa(nxt).cpu().numpy() - ref["next"]).max())
