"""How long does the host take to ISSUE one bench step (4 stream slices), compared with the GPU time of the step?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nimblephysics_amd as na
from util import contact_inputs
dev = "cuda:0"
md, s, a = contact_inputs("atlas20", 4096, 1000, joint_noise=0.002, vel_noise=0.001, action_noise=0.1)
for ns in (1, 4):
    per = 4096 // ns
    worlds = [na.World(md, device=dev) for _ in range(ns)]
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(ns - 1)]
    x = [w.to_soa(torch.tensor(s[i * per:(i + 1) * per], device=dev)) for i, w in enumerate(worlds)]
    u = [w.to_soa(torch.tensor(a[i * per:(i + 1) * per], device=dev)) for i, w in enumerate(worlds)]
    def step():
        for i, (w, st) in enumerate(zip(worlds, streams)):
            with torch.cuda.stream(st):
                w.reset_lcp_cache()
                nxt, sv, status = w.step_soa(x[i], u[i])
                w.backward_soa(sv, 2.0 * nxt)
    for _ in range(20): step()
    torch.cuda.synchronize()
    K = 100
    t0 = time.perf_counter()
    for _ in range(K): step()
    t_issue = (time.perf_counter() - t0) / K
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / K
    print(f"{ns} slices: issue {t_issue*1e3:.3f} ms/step, end-to-end {t_all*1e3:.3f} ms/step")
