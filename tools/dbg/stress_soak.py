"""Stress variants of tools/soak_parity.py (balls mode): dt = 5 ms, 8 x the velocities, body masses scaled by 1e-2 .. 1e2, 200 x the torques; geom: collider sizes x 0.1 .. 5 per axis; mu: friction 1.01e-3 .. 10; tinydt: dt = 1e-5; nograv.
usage (GPU box): python tools/dbg/stress_soak.py <dt|fast|mass|torque|geom|mu|tinydt|nograv|subset|atlimit> <first seed> <count>      (round 2: 150 models each, 0 mismatches)"""
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
    if mode == "geom":            # thin plates, sticks, tiny and big colliders
        for bx in md.boxes[1:]:
            f = tuple(float(10 ** rng.uniform(-1, 0.7)) for _ in range(3))
            bx.size = tuple(x * (f[0] if bx.shape == "sphere" else f[k]) for k, x in enumerate(bx.size))
    if mode == "mu":              # just above the frictionless threshold, and very rough
        for bx in md.boxes:
            bx.mu = float(rng.choice([1.01e-3, 2e-3, 5.0, 10.0]))
    if mode == "tinydt":
        md.dt = 1e-5
    if mode == "subset":          # a random subset of the DOFs is actuated (World::setActionSpace); unmapped torques are zero
        keep = sorted(rng.choice(n, size=max(1, n // 3), replace=False).tolist())
        md.set_action_space(keep); a = a[:, :len(keep)]
    if mode == "atlimit":         # positions, velocities and torques exactly at a joint limit in half of the worlds: clipLossGradientsToBounds
        for i, b in enumerate(md.bodies):
            nd = md.joint_ndof(i)
            if nd == 1 and rng.random() < 0.5:
                b.pos_lo, b.pos_hi = (-0.3,), (0.4,); b.vel_lo, b.vel_hi = (-0.7,), (0.9,); b.force_lo, b.force_hi = (-0.2,), (0.25,)
        md = type(md)(md.name, md.bodies, md.boxes, gravity=md.gravity, dt=md.dt, max_contacts=md.max_contacts)
        fl = md.flat(); s = s.copy(); a = a.copy()
        for d in range(n):
            if np.isfinite(fl["pos_lo"][d]):
                half = rng.random(s.shape[0]) < 0.5
                s[half, d] = rng.choice([fl["pos_lo"][d], fl["pos_hi"][d]], half.sum()); s[half, n + d] = rng.choice([fl["vel_lo"][d], fl["vel_hi"][d]], half.sum())
                a[half, d] = rng.choice([fl["force_lo"][d], fl["force_hi"][d]], half.sum())
    if mode == "nograv":
        md.gravity = (0.0, 0.0, 0.0)
    return md, s, a, g
soak_parity.make_case = make_case
print(mode, soak_parity.run(int(sys.argv[2]), int(sys.argv[3]), 256, verbose=False, balls=True))
