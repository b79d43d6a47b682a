"""Which world of the Jacobian soak is above 1e-5, and does the ORACLE's own dense Jacobian move under 1-ulp perturbations of that state?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import nimblephysics_amd as na
import soak_parity
from oracle import OracleWorld
first, count = int(sys.argv[1]), int(sys.argv[2])
for seed in range(first, first + count):
    case = soak_parity.make_case(seed, 64, balls=True)
    if case is None: continue
    md, s, a, _ = case
    s, a = s[:4], a[:4]
    world = na.World(md, device="cuda:0")
    world.setState(torch.tensor(s)); world.setAction(torch.tensor(a))
    snap = na.neural.forwardPass(world, idempotent=True)
    st = snap.getStatus().cpu().numpy().astype(np.uint32)
    Js = snap.getStateJacobian(world).cpu().numpy()
    ow = OracleWorld(md)
    for b in range(4):
        ow.step(s[b], a[b]); Rs = ow.getStateJacobian()
        e = np.abs(Js[b] - Rs).max() / max(np.abs(Rs).max(), 1e-30)
        if e > 1e-5:
            rng = np.random.default_rng(0)
            spread = 0.0; near = 1e9
            for k in range(32):
                sp = s[b] * (1.0 + rng.integers(-1, 2, s[b].shape) * 2.220446049250313e-16)
                ow.step(sp, a[b]); Rp = ow.getStateJacobian()
                spread = max(spread, np.abs(Rp - Rs).max() / np.abs(Rs).max())
                near = min(near, np.abs(Rp - Js[b]).max() / np.abs(Rs).max())
            print(f"seed {seed} world {b}: err {e:.2e} status {st[b]:#x}/{ow.last_status:#x}; oracle's own spread under 1-ulp perturbations {spread:.2e}, device to nearest outcome {near:.2e}")
print("done")
