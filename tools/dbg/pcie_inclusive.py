"""The PCIe-inclusive rate of the Python surface: timestep() on HOST tensors (the reference's own calling convention: float64 CPU tensors in,
CPU tensors out) - copies to the device, two transposes, the step, the backward pass, copies back.  Never bench.py's `value`.  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from util import contact_inputs
B = 4096
md, s, a = contact_inputs("atlas20", B, 1000, joint_noise=0.02, vel_noise=0.01, action_noise=0.1)
world = na.World(md, device="cuda:0")
for where in ("cpu", "cuda:0"):
    st = torch.tensor(s, device=where, requires_grad=True); at = torch.tensor(a, device=where, requires_grad=True)
    for it in range(3):
        out = timestep(world, st, at); out.pow(2).sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 20
    for it in range(K):
        st.grad = None; at.grad = None
        out = timestep(world, st, at); out.pow(2).sum().backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"timestep() fwd+bwd on {where} tensors [B, 2n] (one World, B = {B}): {dt * 1e3:.3f} ms per step = {B / dt / 1e6:.2f} M worlds*steps/s")
