"""Soak test: long forward rollouts of the standing Atlas and of random ball / box scenes; reports NaNs, status bits and how far
the device drifts from the CPU oracle stepping the same worlds (chaotic contact dynamics amplify round-off, so only the first
steps are expected to agree to 1e-9)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nimblephysics_amd as na
from oracle import OracleWorld
from util import ball_state, ball_world, contact_inputs, box_stack_inputs

def run(name, md, s, a, T, check=8):
    B = s.shape[0]
    w = na.World(md, device="cuda:0")
    x = w.to_soa(torch.tensor(s, device="cuda:0")); u = w.to_soa(torch.tensor(a, device="cuda:0"))
    ow = OracleWorld(md); ref = s[:check].copy()
    bits = np.zeros(B, np.uint32); worst = 0.0
    for t in range(T):
        x, _, st = w.step_soa(x, u, want_saved=True)
        bits |= st.cpu().numpy().astype(np.uint32)
        if t < 20:
            ref = ow.step_batch(ref, a[:check], threads=8)["next"]
            worst = max(worst, float(np.abs(w.from_soa(x)[:check].cpu().numpy() - ref).max()))
    xs = w.from_soa(x).cpu().numpy()
    print(f"{name}: B={B} T={T} finite={np.isfinite(xs).all()} max|q|={np.abs(xs[:, :xs.shape[1]//2]).max():.3g} max|v|={np.abs(xs[:, xs.shape[1]//2:]).max():.3g} "
          f"status bits seen={hex(int(np.bitwise_or.reduce(bits)))} device-oracle over the first 20 steps={worst:.2e}")

md, s, a = contact_inputs("atlas20", 512, 5, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
run("atlas20 standing, no control (falls over)", md, s, a, 600)
md, s, a = box_stack_inputs(512, 6)
run("box stack", md, s, a, 600)
md = ball_world("box_first", n_balls=2)
S, A = zip(*[ball_state(md, [(0.0, 0.0), (0.25 + 0.01 * (i % 7), 0.02 * (i % 5))], i) for i in range(256)])
S = np.array(S); S[:, 6 + 4] += 0.3          # second ball dropped from 0.3 above
run("two balls, one dropped", md, S, 0 * np.array(A), 800)
