"""Experiment: do the latency-bound one-world-per-lane tree kernels of different HIP streams overlap?  (no-contact model)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nimblephysics_amd as na

md = na.atlas("atlas20", ground=False)
n = md.num_dofs
rng = np.random.default_rng(0)
Btot = 4096
s = rng.normal(0, 0.1, (Btot, 2 * n)); a = rng.normal(0, 0.1, (Btot, n)); g = rng.normal(0, 1, (Btot, 2 * n))
dev = "cuda:0"
for nsl in (1, 2, 4, 8):
    Bs = Btot // nsl
    worlds = [na.World(md, device=dev) for _ in range(nsl)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nsl)]
    ins = []
    for i, w in enumerate(worlds):
        sl = slice(i * Bs, (i + 1) * Bs)
        ins.append((w.to_soa(torch.tensor(s[sl], device=dev)), w.to_soa(torch.tensor(a[sl], device=dev)), w.to_soa(torch.tensor(g[sl], device=dev))))
    torch.cuda.synchronize()
    def step():
        for w, st, (x, u, gg) in zip(worlds, streams, ins):
            with torch.cuda.stream(st):
                nxt, saved, status = w.step_soa(x, u)
                w.backward_soa(saved, gg)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t = time.time(); K = 30
    for _ in range(K): step()
    t_issue = (time.time() - t) / K
    torch.cuda.synchronize()
    dt = (time.time() - t) / K
    print(f"slices {nsl}: {dt*1e3:.3f} ms/step (cpu issue {t_issue*1e3:.3f} ms)  {Btot/dt/1e6:.2f} M worlds*steps/s", flush=True)
