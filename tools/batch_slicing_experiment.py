"""Experiment: slice the batch over several HIP streams (one World per slice) — do the kernels of different slices overlap?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nimblephysics_amd as na
from util import contact_inputs

md, s, a = contact_inputs("atlas20", 4096, 1)
g = np.random.default_rng(5).normal(0, 1, s.shape)
dev = "cuda:0"
pool = [torch.cuda.Stream(device=dev) for _ in range(4)]
for st in pool:
    with torch.cuda.stream(st):
        torch.cuda._sleep(1000000)
torch.cuda.synchronize()
for nsl in (1, 2, 4, 1, 2, 4):
    Bs = 4096 // nsl
    worlds = [na.World(md, device=dev) for _ in range(nsl)]
    ins = []
    for i, w in enumerate(worlds):
        sl = slice(i * Bs, (i + 1) * Bs)
        ins.append((w.to_soa(torch.tensor(s[sl], device=dev)), w.to_soa(torch.tensor(a[sl], device=dev)), w.to_soa(torch.tensor(g[sl], device=dev))))
    torch.cuda.synchronize()
    def step():
        for w, st, (x, u, gg) in zip(worlds, pool, ins):
            with torch.cuda.stream(st):
                w.reset_lcp_cache()
                nxt, saved, status = w.step_soa(x, u)
                w.backward_soa(saved, gg)
    for _ in range(25): step()
    torch.cuda.synchronize()
    t = time.time(); K = 40
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.time() - t) / K
    print(f"slices {nsl}: {dt*1e3:.3f} ms/step  {4096/dt/1e6:.2f} M worlds*steps/s", flush=True)
