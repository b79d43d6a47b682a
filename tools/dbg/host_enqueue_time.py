"""Developer check: how long does the HOST take to enqueue one fwd+bwd step of the default bench (four stream slices, the Python path of
bench.py) against how long the GPU takes to run it?  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from util import contact_inputs
dev = torch.device("cuda:0")
B, S = 4096, 4
md, s, a = contact_inputs("atlas20", B, 1000, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
bounds = [(i * B // S, (i + 1) * B // S) for i in range(S)]
worlds = [na.World(md, device=dev) for _ in bounds]
streams = [torch.cuda.Stream(dev) for _ in bounds]
st0 = [w.to_soa(torch.tensor(s[lo:hi], device=dev)) for w, (lo, hi) in zip(worlds, bounds)]
ac = [w.to_soa(torch.tensor(a[lo:hi], device=dev)) for w, (lo, hi) in zip(worlds, bounds)]
k = worlds[0].k
def run(T):
    ga_total = [torch.zeros((k, hi - lo), dtype=torch.float64, device=dev) for (lo, hi) in bounds]
    for _ in range(T):
        for i, (w, stt) in enumerate(zip(worlds, streams)):
            with torch.cuda.stream(stt):
                w.reset_lcp_cache()
                nxt, sv, status = w.step_soa(st0[i], ac[i], want_saved=True)
                gs, ga = w.backward_soa(sv, 2.0 * nxt)
                ga_total[i] += ga
for w in worlds: w.set_timing(False)
run(8); torch.cuda.synchronize()
for T in (20, 64):
    for rep in range(3):
        t0 = time.perf_counter(); run(T); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"T={T}: host enqueue {1e3 * (t1 - t0) / T:.3f} ms/step, until the GPU is done {1e3 * (t2 - t0) / T:.3f} ms/step")
