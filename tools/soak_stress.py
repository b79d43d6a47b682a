"""Stress variants of the randomised parity soak (tools/soak_parity.py, ball-joint mode): the same random models pushed to the edges of
the parameter space, every world's next state and both gradients against the oracle with the soak's criterion.  In the GPU suite at
reduced size as tests/test_gpu_stress.py.
  dt      time step 5 ms                       tinydt  time step 1e-5
  fast    8 x the velocities                   torque  200 x the torques
  mass    body masses scaled by 1e-2 .. 1e2    nograv  no gravity
  geom    collider sizes x 0.1 .. 5 per axis (thin plates, sticks, tiny and big colliders)
  mu      friction from 1.01e-3 (just above the frictionless threshold) to 10
  subset  a random third of the DOFs actuated (World::setActionSpace; unmapped torques are zero)
  atlimit positions, velocities and torques exactly at their limits in half of the worlds (clipLossGradientsToBounds)
  limits  like atlimit, and those joints ENFORCE their position limits (Joint::setPositionLimitEnforced): joint-limit rows in the LCP next to
          the contact rows (JointLimitConstraint.cpp), half of the limited DOFs exactly at a limit, a quarter beyond it
  selfcol every skeleton checks self-collisions (Skeleton::enableSelfCollisionCheck), joint angles x 3 so that limbs fold onto each other:
          contacts between two bodies of one tree, DOFs above both of them.  (Without the adjacent-body check: two bodies joined by ONE
          single-DOF joint have a rank-1 Delassus block, and the reference's stage 0 then succeeds or fails with the last bit of A -
          1e-15 of noise on the oracle's own A flips it, tools/dbg notes in DESIGN.md section 5 - which no perturbation of the STATE probes.)
  adjacent like selfcol, and half of the skeletons also check a body against its parent (enableAdjacentBodyCheck)
  capsule every box collider becomes a capsule (radius = half its smallest side, cylinder height = its longest side, axis = that side's) and
          the ground a world-fixed sphere of radius 100 m with its top at y = 0 (a capsule cannot meet a box: libccd in the reference)
  mix     a random subset (each with probability 1/2, drawn from the seed) of capsule, geom, mass, mu, selfcol, limits, subset, dt, fast,
          torque, nograv in ONE
          model: the interactions of the features (limit rows next to capsule and self-collision contacts in the eight slots, ...)
  a+b+c   the named mutations one after the other
usage (GPU box): python tools/soak_stress.py <mode> [first seed] [count] [B] [balls|big|multi]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))

MODES = ("dt", "tinydt", "fast", "torque", "mass", "nograv", "geom", "mu", "subset", "atlimit", "capsule", "limits", "selfcol", "adjacent")


MIX_ORDER = ("capsule", "geom", "mass", "mu", "selfcol", "limits", "subset", "dt", "fast", "torque", "nograv")   # (limits rebuilds the description: before subset)


def mutator(mode, slots=None):
    if mode == "mix" or "+" in mode:
        def chain(seed, md, s, a, g):
            if mode == "mix":
                pick = np.random.default_rng(seed + 77).random(len(MIX_ORDER)) < 0.5          # (the first eight draws are the ones of the eight-mode mix)
                parts = [m for m, p in zip(MIX_ORDER, pick) if p]
            else:
                parts = mode.split("+")
            for k, m in enumerate(parts):
                md, s, a, g = mutator(m, slots)(seed + 1000003 * k, md, s, a, g)
            return md, s, a, g
        return chain
    assert mode in MODES, mode

    def mutate(seed, md, s, a, g):
        rng = np.random.default_rng(seed)
        n = md.num_dofs
        if mode == "dt":
            md.dt = 5e-3
        elif mode == "tinydt":
            md.dt = 1e-5
        elif mode == "fast":
            s = s.copy(); s[:, n:] *= 8.0
        elif mode == "torque":
            a = a * 200.0
        elif mode == "mass":
            for b in md.bodies:
                f = float(10 ** rng.uniform(-2, 2)); b.mass *= f; b.inertia = tuple(x * f for x in b.inertia)
        elif mode == "nograv":
            md.gravity = (0.0, 0.0, 0.0)
        elif mode == "geom":
            for bx in md.boxes[1:]:
                f = tuple(float(10 ** rng.uniform(-1, 0.7)) for _ in range(3))
                bx.size = tuple(x * (f[0] if bx.shape == "sphere" else f[k]) for k, x in enumerate(bx.size))
        elif mode == "mu":
            for bx in md.boxes:
                bx.mu = float(rng.choice([1.01e-3, 2e-3, 5.0, 10.0]))
        elif mode == "subset":
            keep = sorted(rng.choice(n, size=max(1, n // 3), replace=False).tolist())
            md.set_action_space(keep); a = a[:, :len(keep)]
        elif mode in ("selfcol", "adjacent"):
            # adjacent: ... and, skeleton by skeleton with probability 1/2, also between a body and its parent (Skeleton::enableAdjacentBodyCheck):
            # rank-1 Delassus blocks, judged by the rounding-level probes and the replay of tools/soak_parity.py
            sk = md.body_skeletons() if hasattr(md, "body_skeletons") else [0] * len(md.bodies)
            adj = {k: bool(rng.random() < 0.5) for k in sorted(set(sk))}
            for i, b in enumerate(md.bodies):
                b.self_collision = True
                if mode == "adjacent" and adj[sk[i]]:
                    b.adjacent_body_check = True
            # (only the single-DOF joints: exponential coordinates near pi are where the ORACLE's finite-differenced integration Jacobian
            #  loses its digits)
            s = s.copy()
            off = 0
            for i in range(len(md.bodies)):
                nd = md.joint_ndof(i)
                if nd == 1:
                    s[:, off] *= 3.0
                off += nd
        elif mode == "capsule":
            import nimblephysics_amd as na
            g0 = md.boxes[0]
            assert g0.body < 0 and g0.shape == "box"
            ground = na.SphereSpec(-1, na.make_transform((0.0, -100.0, 0.0)), 100.0, g0.mu)
            ground.restitution = g0.restitution
            md.boxes[0] = ground
            for bx in md.boxes[1:]:
                if bx.shape != "box":
                    continue
                size = np.asarray(bx.size, dtype=np.float64)
                ax = int(np.argmax(size))
                Rz = np.eye(4)                                    # the capsule's z axis along the box's longest side
                if ax == 0:
                    Rz[:3, :3] = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], dtype=np.float64)
                elif ax == 1:
                    Rz[:3, :3] = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=np.float64)
                bx.T = np.asarray(bx.T, dtype=np.float64) @ Rz
                r = float(0.5 * np.min(size))
                bx.size = (r, float(max(size[ax] - 2 * r, 0.0)), 0.0)
                bx.shape = "capsule"
        elif mode in ("atlimit", "limits"):
            for i, b in enumerate(md.bodies):
                if md.joint_ndof(i) == 1 and rng.random() < (0.5 if mode == "atlimit" else 0.35):
                    b.pos_lo, b.pos_hi = (-0.3,), (0.4,); b.vel_lo, b.vel_hi = (-0.7,), (0.9,); b.force_lo, b.force_hi = (-0.2,), (0.25,)
                    b.limit_enforced = mode == "limits"
                elif mode == "limits" and md.joint_ndof(i) == 3 and b.joint_type == "ball" and rng.random() < 0.35:
                    # limits on the exponential coordinates of a ball joint (JointLimitConstraint works on any joint's coordinates)
                    b.pos_lo, b.pos_hi = (-0.3, -0.25, -0.35), (0.4, 0.3, 0.25); b.limit_enforced = True
            # (enforced limits share the constraint slots with the contacts: 16 slots - the 48-row build - so that no world is truncated)
            md = type(md)(md.name, md.bodies, md.boxes, gravity=md.gravity, dt=md.dt, max_contacts=(slots or 16) if mode == "limits" else md.max_contacts)
            fl = md.flat(); s = s.copy(); a = a.copy()
            for d in range(n):
                if np.isfinite(fl["pos_lo"][d]):
                    half = rng.random(s.shape[0]) < 0.5
                    s[half, d] = rng.choice([fl["pos_lo"][d], fl["pos_hi"][d]], half.sum())
                    if mode == "limits":                       # ... and some of them beyond the limit
                        s[half, d] += rng.choice([0.0, 0.0, -0.02, 0.02], half.sum()) * (np.abs(s[half, d]) > 0)
                    if np.isfinite(fl["vel_lo"][d]) and np.isfinite(fl["vel_hi"][d]):      # (a ball joint's coordinates carry position limits only)
                        s[half, n + d] = rng.choice([fl["vel_lo"][d], fl["vel_hi"][d]], half.sum())
                    if np.isfinite(fl["force_lo"][d]) and np.isfinite(fl["force_hi"][d]):
                        a[half, d] = rng.choice([fl["force_lo"][d], fl["force_hi"][d]], half.sum())
        return md, s, a, g
    return mutate


def run(mode, first=0, count=20, B=256, verbose=False, variant="balls", slots=None):
    """variant: the model family of tools/soak_parity.py the mutation is applied to (balls, big, multi)."""
    import soak_parity
    return soak_parity.run(first, count, B, verbose=verbose, balls=variant == "balls", big=variant == "big", multi=variant == "multi",
                           mutate=mutator(mode, slots), slots=slots)


if __name__ == "__main__":
    print(sys.argv[1], run(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 150,
                           int(sys.argv[4]) if len(sys.argv) > 4 else 256, variant=sys.argv[5] if len(sys.argv) > 5 else "balls"))
