"""Stress variants of tools/soak_parity.py (balls mode): dt = 5 ms, 8 x the velocities, body masses scaled by 1e-2 .. 1e2, 200 x the torques.
usage (GPU box): python tools/dbg/stress_soak.py <dt|fast|mass|torque> <first seed> <count>      (round 2: 150 models each, 0 mismatches)"""
import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import soak_parity
orig = soak_parity.make_case
mode = sys.argv[1]
def make_case(seed, B=256, big=False, multi=False, balls=False, far=False):
    c = orig(seed, B, big, multi, balls, far)
    if c is None: return None
    md, s, a, g = c
    rng = np.random.default_rng(seed)
    n = md.num_dofs
    if mode == "dt":
        md.dt = 5e-3
    if mode == "fast":
        s = s.copy(); s[:, n:] *= 8.0
    if mode == "mass":
        for b in md.bodies:
            f = float(10 ** rng.uniform(-2, 2)); b.mass *= f; b.inertia = tuple(x * f for x in b.inertia)
    if mode == "torque":
        a = a * 200.0
    return md, s, a, g
soak_parity.make_case = make_case
print(mode, soak_parity.run(int(sys.argv[2]), int(sys.argv[3]), 256, verbose=False, balls=True))
