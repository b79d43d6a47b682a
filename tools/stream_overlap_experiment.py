"""Do the library's kernels overlap across HIP streams?  (after warming the streams up)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nimblephysics_amd as na
md = na.atlas("atlas20", ground=False)
n = md.num_dofs
rng = np.random.default_rng(0)
dev = "cuda:0"
def make(B):
    w = na.World(md, device=dev)
    s = w.to_soa(torch.tensor(rng.normal(0, 0.1, (B, 2 * n)), device=dev)); a = w.to_soa(torch.tensor(rng.normal(0, 0.1, (B, n)), device=dev))
    return w, s, a
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
for st in streams:
    with torch.cuda.stream(st):
        torch.cuda._sleep(1000000)
torch.cuda.synchronize()
for nsl, B in ((1, 4096), (1, 2048), (2, 2048), (4, 1024), (4, 4096)):
    ws = [make(B) for _ in range(nsl)]
    def step():
        for (w, s, a), st in zip(ws, streams):
            with torch.cuda.stream(st):
                w.step_soa(s, a, want_saved=False)
    for _ in range(20): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50): step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 50
    print(f"{nsl} streams x B={B}: {dt*1e3:.3f} ms per round of forward kernels", flush=True)
