"""Seeded synthetic inputs for the BASELINE.md configs (numpy default_rng(seed))."""
import numpy as np

import nimblephysics_amd as na


def rel_err(x, y):
    x, y = np.asarray(x), np.asarray(y)
    return float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-30))


def cfg_inputs(name, B, seed):
    """Returns (model_description, state[B,2n], action[B,k])."""
    rng = np.random.default_rng(seed)
    if name == "pendulum":  # cfg1
        md = na.single_pendulum()
        q = rng.uniform(-np.pi, np.pi, (B, 1)); v = rng.uniform(-1, 1, (B, 1)); a = rng.uniform(-1, 1, (B, 1))
    elif name == "cartpole":  # cfg2
        md = na.cartpole()
        q = np.stack([rng.uniform(-1, 1, B), rng.uniform(-np.pi / 2, np.pi / 2, B)], 1)
        v = rng.normal(0, 1, (B, 2))
        a = np.stack([rng.uniform(-10, 10, B), np.zeros(B)], 1)
    elif name in ("atlas33", "atlas20"):  # cfg3 (free fall, no ground)
        md = na.atlas(name)
        fl = md.merge_welds().flat()
        n = md.num_dofs
        q = np.zeros((B, n))
        q[:, 0:3] = rng.normal(0, 0.3, (B, 3)); q[:, 0] += -np.pi / 2
        lo, hi = fl["pos_lo"][6:], fl["pos_hi"][6:]
        q[:, 6:] = rng.uniform(0.2 * lo, 0.2 * hi, (B, n - 6))
        v = rng.normal(0, 0.1, (B, n))
        a = rng.normal(0, 1, (B, n))
    else:
        raise KeyError(name)
    return md, np.concatenate([q, v], 1), a


def contact_inputs(name, B, seed, joint_noise=0.002, vel_noise=0.001, action_noise=0.1):
    """Standing Atlas on the ground box (cfg5 / metric config pose: q[0] = -pi/2, q[4] = -0.01,
    unittests/unit/test_AtlasGradients.cpp:235-236) with small joint noise: both feet flat, 8 foot-corner
    contacts, contact stays in the sticking regime where the LCP warm start / guess is already valid."""
    rng = np.random.default_rng(seed)
    md = na.atlas(name, ground=True)
    n = md.num_dofs
    q = np.zeros((B, n)); q[:, 0] = -np.pi / 2; q[:, 4] = -0.01
    q[:, 6:] = rng.normal(0, joint_noise, (B, n - 6))
    v = rng.normal(0, vel_noise, (B, n))
    a = rng.normal(0, action_noise, (B, n))
    return md, np.concatenate([q, v], 1), a
