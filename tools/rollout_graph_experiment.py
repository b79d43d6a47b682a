"""does replaying a T = 64 rollout (fwd + bwd) from ONE captured HIP graph beat the eager launches of the host loop?
Atlas-33 on the ground, B = 8192 (cfg5's per-GPU share), warm-started rollout, the library's own slicing inside the call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nimblephysics_amd as na
import bench

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = sys.argv[3] if len(sys.argv) > 3 else "atlas33_contact"
md, s, a, desc = bench.make_workload(wl, B, 1000, 0.02)
w = na.World(md, device=dev)
x0 = w.to_soa(torch.tensor(s, device=dev)); u = w.to_soa(torch.tensor(a, device=dev))
res = {}


def pass_():
    states, sv, st = w.rollout_soa(x0, u, T=T, want_saved=True, warm_start=True)
    gst = torch.zeros_like(states); gst[-1] = 2.0 * states[-1]
    g0, ga = w.rollout_backward_soa(sv, gst)
    res["g0"], res["ga"], res["last"] = g0, ga, states[-1]


cap = torch.cuda.Stream(device=dev)
with torch.cuda.stream(cap):
    for _ in range(6):
        pass_()
torch.cuda.synchronize()
eager = {k: v.clone() for k, v in res.items()}
K = 12
t0 = time.perf_counter()
with torch.cuda.stream(cap):
    for _ in range(K):
        pass_()
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / K
print(f"eager: {te*1e3:.2f} ms per rollout fwd+bwd = {B*T/te/1e6:.2f} M world-steps/s", flush=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=cap):
        pass_()
except Exception as e:
    print("capture failed:", repr(e)[:400]); sys.exit(0)
torch.cuda.synchronize()
out = dict(res)
for _ in range(4):
    g.replay()
torch.cuda.synchronize()
same = all(torch.equal(out[k], eager[k]) for k in eager)
t0 = time.perf_counter()
for _ in range(K):
    g.replay()
torch.cuda.synchronize()
tg = (time.perf_counter() - t0) / K
print(f"graph: {tg*1e3:.2f} ms per rollout fwd+bwd = {B*T/tg/1e6:.2f} M world-steps/s; replay reproduces the eager result bit for bit: {same}")
