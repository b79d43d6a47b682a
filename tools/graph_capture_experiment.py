"""Can a forward+backward step be captured in a HIP graph (torch.cuda.CUDAGraph) and does replaying it beat eager launches?
One World per slice, every slice on its own stream inside the capture (fork / join through the capture stream)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nimblephysics_amd as na
from util import contact_inputs
dev = "cuda:0"
B = 4096
md, s, a = contact_inputs("atlas20", B, 1000, joint_noise=0.002, vel_noise=0.001, action_noise=0.1)
for ns in (1, 2, 4, 8):
    per = B // ns
    worlds = [na.World(md, device=dev) for _ in range(ns)]
    x = [w.to_soa(torch.tensor(s[i * per:(i + 1) * per], device=dev)) for i, w in enumerate(worlds)]
    u = [w.to_soa(torch.tensor(a[i * per:(i + 1) * per], device=dev)) for i, w in enumerate(worlds)]
    side = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    outs = [None] * ns

    def step_all(main):
        for i, (w, st) in enumerate(zip(worlds, side)):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                w.reset_lcp_cache()
                nxt, sv, status = w.step_soa(x[i], u[i])
                gs, ga = w.backward_soa(sv, 2.0 * nxt)
                outs[i] = (nxt, gs, ga)
        for st in side:
            main.wait_stream(st)

    cap = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(cap):
        for _ in range(3):
            step_all(cap)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=cap):
            step_all(cap)
    except Exception as e:
        print(ns, "capture failed:", repr(e)[:300]); continue
    torch.cuda.synchronize()
    graph_outs = outs[:]                       # the tensors the captured kernels write into
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    got = [o[1].clone() for o in graph_outs]
    with torch.cuda.stream(cap):
        step_all(cap)                          # eager reference (fresh output tensors)
    torch.cuda.synchronize()
    same = all(torch.equal(o[1], r) for o, r in zip(outs, got))
    K = 64
    t0 = time.perf_counter()
    for _ in range(K):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    with torch.cuda.stream(cap):
        for _ in range(K):
            step_all(cap)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / K
    print(f"{ns} slices: graph {tg*1e3:.3f} ms/step ({B/tg/1e6:.2f} M/s), eager {te*1e3:.3f} ms/step ({B/te/1e6:.2f} M/s), replay reproduces: {same}")
