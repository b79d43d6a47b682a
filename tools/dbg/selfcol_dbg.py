import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import nimblephysics_amd as na
import soak_parity, soak_stress
from oracle import OracleWorld
seed = int(sys.argv[1])
md, s, a, g = soak_parity.make_case(seed, 256, balls=True)
md, s, a, g = soak_stress.mutator("selfcol")(seed, md, s, a, g)
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
nxt, saved, status = world.step_soa(st, at)
status = status.cpu().numpy().astype(np.uint32)
ref = ow.step_batch(s, a, None, threads=8)
bad = np.where((status & 0x481) != (ref["status"] & 0x481))[0]
print("bodies", [(i, b.name, b.joint_type, b.parent) for i, b in enumerate(md.bodies)])
print("boxes", [(bx.body, bx.shape) for bx in md.boxes], "adjacent flag", md.bodies[0].adjacent_body_check)
print("bad worlds", bad[:10], [hex(x) for x in status[bad[:10]]], [hex(x) for x in ref["status"][bad[:10]]])
for wd in bad[:3]:
    ow.reset_lcp_cache(); ow.step(s[wd], a[wd]); c = ow.last_contacts()
    print(wd, "oracle contacts", c.shape[0], c[:, 8:10].astype(int).tolist(), c[:, 6])
