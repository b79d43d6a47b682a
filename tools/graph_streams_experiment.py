"""One HIP graph per batch slice (a LINEAR chain: forward + loss gradient + backward of that slice), each replayed on its own stream,
against the eager launches of bench.py.  usage (GPU box): python tools/graph_streams_experiment.py [noise]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.graph import GraphedStep
from util import contact_inputs

noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
B, steps = 4096, 64
dev = torch.device("cuda", 0)
md, s, a = contact_inputs("atlas20", B, 1000, joint_noise=noise, vel_noise=noise / 2, action_noise=0.1)


def run(nsl, graphs):
    per = B // nsl
    worlds = [na.World(md, device=dev) for _ in range(nsl)]
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(nsl - 1)]   # at most 4 streams in flight: a fifth costs 2x (hardware queues)
    st = [w.to_soa(torch.tensor(s[i * per:(i + 1) * per], device=dev)) for i, w in enumerate(worlds)]
    at = [w.to_soa(torch.tensor(a[i * per:(i + 1) * per], device=dev)) for i, w in enumerate(worlds)]
    acc = [torch.zeros((worlds[0].k, per), dtype=torch.float64, device=dev) for _ in range(nsl)]
    torch.cuda.synchronize()
    if graphs:
        gsteps = []
        for i, w in enumerate(worlds):
            g = GraphedStep(w, per, loss_grad=lambda nxt: 2.0 * nxt)
            g.state.copy_(st[i]); g.action.copy_(at[i])
            g.capture()
            gsteps.append(g)
        torch.cuda.synchronize()

        def step():
            for i, g in enumerate(gsteps):
                with torch.cuda.stream(streams[i]):
                    g._graph.replay()
                    acc[i] += g.grad_action
    else:
        def step():
            for i, w in enumerate(worlds):
                with torch.cuda.stream(streams[i]):
                    w.reset_lcp_cache()
                    nxt, sv, _ = w.step_soa(st[i], at[i])
                    gs, ga = w.backward_soa(sv, 2.0 * nxt)
                    acc[i] += ga
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"slices {nsl} {'graphs' if graphs else 'eager '}: {dt * 1e3:.3f} ms/step  {B / dt / 1e6:.2f} M worlds*steps/s", flush=True)


for nsl, graphs in ((4, False), (4, True), (2, True), (8, True), (16, True)):
    try:
        run(nsl, graphs)
    except Exception as e:
        print("slices", nsl, "graphs", graphs, "failed:", repr(e)[:200])
