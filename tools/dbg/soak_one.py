import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import nimblephysics_amd as na
from oracle import OracleWorld
from test_gpu_random_trees import random_tree
seed = int(sys.argv[1]); big = len(sys.argv) > 2
B = 256
rng = np.random.default_rng(50000 + seed)
nb = int(rng.integers(8, 22)) if big else int(rng.integers(1, 10))
md = random_tree(rng, nb, rng.choice(["chain", "star", "random"]), True, welds=0.2 if rng.random() < 0.3 else 0,
                 colliders=int(rng.integers(3, 8)) if big else int(rng.integers(1, 4)), spheres=bool(rng.random() < 0.4))
for bx in md.boxes:
    r = rng.random()
    bx.mu = 0.0 if r < 0.15 else (float(rng.uniform(0.05, 1.5)))
    bx.restitution = float(rng.uniform(0.3, 1.0)) if rng.random() < 0.3 else 0.0
md.penetration_correction = bool(rng.random() < 0.3)
n = md.num_dofs
q = rng.normal(0, 0.25, (B, n)); q[:, 3] = rng.normal(0, 0.3, B); q[:, 5] = rng.normal(0, 0.3, B)
q[:, 4] = rng.uniform(0.02, 0.5, B)
v = rng.normal(0, rng.choice([0.05, 0.5, 2.0]), (B, n))
s = np.concatenate([q, v], 1); a = rng.normal(0, 0.5, (B, len(md.action_map)))
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
nxt, saved, status = world.step_soa(st, at)
status = status.cpu().numpy().astype(np.uint32)
ref = ow.step_batch(s, a, None, threads=8)
d = np.where((status & 0x81) != (ref["status"] & 0x81))[0]
print("nb", nb, "n", n, "boxes", [(bx.shape, bx.body) for bx in md.boxes], "welds", md.has_welds())
print("differ:", d, [hex(x) for x in status[d]], [hex(x) for x in ref["status"][d]])
np.set_printoptions(linewidth=200, precision=6, suppress=False)
lay_nc = 5 * n
sv = saved.view(torch.float64).cpu().numpy()
for wd in d[:3]:
    ow.step(s[wd], a[wd])
    cts = ow.last_contacts()
    print("oracle: n contacts", len(cts))
    for c in cts: print("   p", c[0:3], "n", c[3:6], "depth", c[6], "type", c[7], "rest", c[8:12])
    rows = sv[: (5 * n + 1 + 8 * 22) * B].reshape(-1, B)
    nc = int(rows[lay_nc, wd])
    print("device: nc", nc)
    for k in range(nc):
        r = rows[lay_nc + 1 + k * 22: lay_nc + 1 + (k + 1) * 22, wd]
        print("   p", r[0:3], "n", r[3:6], "depth", r[6], "type", r[7], "boxA", r[8], "boxB", r[9])
