import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, copy
import nimblephysics_amd as na
import soak_parity, soak_stress
from oracle import OracleWorld
seed, wd = int(sys.argv[1]), int(sys.argv[2])
md, s, a, g = soak_parity.make_case(seed, 256, balls=True)
md, s, a, g = soak_stress.mutator("selfcol")(seed, md, s, a, g)
def run(m, tag):
    world = na.World(m, device="cuda:0")
    st = world.to_soa(torch.tensor(s[wd:wd + 1], device="cuda:0")); at = world.to_soa(torch.tensor(a[wd:wd + 1], device="cuda:0"))
    nxt, saved, status = world.step_soa(st, at)
    ow = OracleWorld(m); ow.step(s[wd], a[wd])
    print(tag, "dev", hex(int(status[0])), "oracle", hex(ow.last_status), "contacts", ow.last_contacts()[:, 6:10].tolist())
run(md, "as is")
m2 = copy.deepcopy(md)
for b in m2.bodies: b.self_collision = False
run(m2, "self-collision off")
m3 = copy.deepcopy(md); m3.boxes = m3.boxes[1:]
run(m3, "no ground")
m4 = copy.deepcopy(md)
for b in m4.bodies: b.adjacent_body_check = False
run(m4, "adjacent off")
print("q", s[wd, :md.num_dofs])
