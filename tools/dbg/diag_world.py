import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nimblephysics_amd as na
from oracle import OracleWorld
from util import box_stack_inputs
np.set_printoptions(linewidth=250, precision=5)
B, seed, wd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mus = [float(x) for x in sys.argv[4:7]] if len(sys.argv) > 4 else None
md, s, a = box_stack_inputs(B, seed)
if mus:
    for bx, mu in zip(md.boxes, mus): bx.mu = mu
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
g = np.random.default_rng(seed + 1).normal(0, 1, s.shape)
n = 12
st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
nxt, saved, status = world.step_soa(st, at)
gs, ga = world.backward_soa(saved, world.to_soa(torch.tensor(g, device="cuda:0")))
torch.cuda.synchronize()
sv = saved.view(torch.float64).cpu().numpy()
row = lambda r: sv[r * B + wd]
x0 = 5 * n + 1 + 8 * 22
X = np.array([row(x0 + i) for i in range(24)]); bb = np.array([row(x0 + 24 + i) for i in range(24)]); cls = np.array([row(x0 + 48 + i) for i in range(24)])
print("dev status", hex(int(status[wd])), "nc", row(5 * n), "cfm", row(x0 + 72), "pflag", row(x0 + 73))
print("dev X  ", X); print("dev b  ", bb); print("dev cls", cls)
ow.reset_lcp_cache(); nx = ow.step(s[wd], a[wd]); ogs, oga = ow.backprop(g[wd])
l = ow.last_lcp()
print("ora status", hex(ow.last_status)); print("ora X  ", l["x"]); print("ora b  ", l["b"]); print("ora cls", l["row_class"]); print("ora fidx", l["findex"])
print("contacts types", ow.last_contacts()[:, 7], "depth", ow.last_contacts()[:, 6])
dgs = gs[:, wd].cpu().numpy()
print("next err", np.abs(nxt[:, wd].cpu().numpy() - nx).max())
print("grad dev", dgs); print("grad ora", ogs); print("diff    ", dgs - ogs)
