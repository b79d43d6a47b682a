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


def box_stack_inputs(B, seed, overhang=False):
    """cfg4: welded ground box + two FreeJoint cubes (0.2^3, mu = 1).  Default: box 2 stacked on box 1 with the same
    yaw and a small lateral offset -> 4 + 4 = 8 contacts (FACE_VERTEX below, EDGE_EDGE / FACE_VERTEX / VERTEX_FACE
    between the cubes), penetration 1e-4..1e-3.  overhang=True: one cube hanging over the rim of the ground box
    (EDGE_EDGE contacts against the world-fixed box, stage-0 regime), the other cube parked far above."""
    rng = np.random.default_rng(seed)
    md = na.box_stack()
    # geometry of the loaded model: top face of the ground box and the zero-configuration centres of the two cubes
    gb = md.boxes[0]
    top = (md.bodies[0].T_pj @ gb.T)[1, 3] + 0.5 * gb.size[1]
    c1, c2 = md.bodies[1].T_pj[:3, 3], md.bodies[2].T_pj[:3, 3]
    q = np.zeros((B, 12)); v = np.zeros((B, 12))
    pen1, pen2 = rng.uniform(1e-4, 1e-3, B), rng.uniform(1e-4, 1e-3, B)
    yaw = rng.uniform(-np.pi, np.pi, B)
    if overhang:
        q[:, 1] = yaw; q[:, 3] = rng.uniform(0.9, 0.99, B) * rng.choice([-1, 1], B) - c1[0]; q[:, 4] = top + 0.1 - pen1 - c1[1]; q[:, 5] = rng.uniform(-0.5, 0.5, B) - c1[2]
        q[:, 10] = 5.0
        v[:, 0:6] = rng.normal(0, 0.001, (B, 6))
    else:
        x1, z1 = rng.uniform(-0.3, 0.3, B), rng.uniform(-0.3, 0.3, B)
        q[:, 1] = yaw; q[:, 4] = top + 0.1 - pen1 - c1[1]; q[:, 3] = x1 - c1[0]; q[:, 5] = z1 - c1[2]
        off = rng.uniform(0.005, 0.03, (B, 2)) * rng.choice([-1, 1], (B, 2))
        c, s_ = np.cos(yaw), np.sin(yaw)
        q[:, 7] = yaw
        q[:, 9] = x1 + c * off[:, 0] + s_ * off[:, 1] - c2[0]; q[:, 11] = z1 - s_ * off[:, 0] + c * off[:, 1] - c2[2]
        q[:, 10] = top + 0.3 - pen1 - pen2 - c2[1]
        v[:, [3, 5, 9, 11]] = rng.normal(0, 0.05, (B, 4))
    a = np.zeros((B, 12))
    return md, np.concatenate([q, v], 1), a


# ---- sphere colliders (tests/test_oracle_spheres.py, tests/test_gpu_spheres.py) ----
def ball_world(order="box_first", n_balls=1, radius=0.1, arm=False):
    """Free-joint balls over a welded ground box (top face at y = 0)."""
    bodies, cols = [], []
    ground = na.BoxSpec(-1, na.make_transform((0, -0.5, 0)), (4.0, 1.0, 4.0), 1.0)
    for i in range(n_balls):
        I = 0.4 * 1.0 * radius * radius
        bodies.append(na.BodySpec(f"ball{i}", -1, "free", f"ball{i}_joint", mass=1.0, inertia=(I, I, I, 0, 0, 0)))
        cols.append(na.SphereSpec(i, np.eye(4), radius, 0.8))
    if arm:   # a revolute link carrying a second sphere, hinged on ball 0
        bodies.append(na.BodySpec("arm", 0, "revolute", "arm_joint", axis=(0, 0, 1), T_pj=na.make_transform((0.15, 0, 0)),
                                  T_cj=na.make_transform((-0.15, 0, 0)), mass=0.5, inertia=(0.002, 0.002, 0.002, 0, 0, 0)))
        cols.append(na.SphereSpec(len(bodies) - 1, np.eye(4), radius, 0.8))
    boxes = [ground] + cols if order == "box_first" else cols + [ground]
    return na.ModelDescription("balls", bodies, boxes, max_contacts=8)


# ---- self-collision (tests/test_oracle_self_collision.py, tests/test_gpu_self_collision.py) ----
def folding_arm(self_collision=True, tip="sphere", adjacent=False):
    """A planar three-link arm (revolute joints about z, links of 0.3 m along their x axes, a box on each of the first two links) whose last
    link folds back onto the FIRST one (q1 ~ 2.1, q2 ~ 1.9): a contact between two non-adjacent bodies of one skeleton, with joint 0 above
    both of them.  tip: the collider of the last link, "sphere" or "box"."""
    I = (0.002, 0.002, 0.002, 0, 0, 0)
    kw = dict(self_collision=self_collision, adjacent_body_check=adjacent)
    bodies = [na.BodySpec("l0", -1, "revolute", "j0", axis=(0, 0, 1), T_cj=na.make_transform((-0.15, 0, 0)), mass=1.0, inertia=I, **kw),
              na.BodySpec("l1", 0, "revolute", "j1", axis=(0, 0, 1), T_pj=na.make_transform((0.15, 0, 0)), T_cj=na.make_transform((-0.15, 0, 0)), mass=0.8, inertia=I, **kw),
              na.BodySpec("l2", 1, "revolute", "j2", axis=(0, 0, 1), T_pj=na.make_transform((0.15, 0, 0)), T_cj=na.make_transform((-0.15, 0, 0)), mass=0.6, inertia=I, **kw)]
    cols = [na.BoxSpec(0, np.eye(4), (0.3, 0.06, 0.06), 0.8), na.BoxSpec(1, np.eye(4), (0.3, 0.06, 0.06), 0.8),
            na.SphereSpec(2, na.make_transform((0.12, 0, 0)), 0.04, 0.8) if tip == "sphere" else na.BoxSpec(2, na.make_transform((0.1, 0, 0), rpy=(0.5, 0.3, 0.2)), (0.1, 0.06, 0.06), 0.8)]   # (tilted: no parallel edges)
    return na.ModelDescription("folding_arm", bodies, cols, gravity=(0, -9.81, 0), max_contacts=8)


# ---- joint-limit rows (tests/test_oracle_joint_limits.py, tests/test_gpu_joint_limits.py) ----
def limited_arm(enforce=True, ground=False, n_links=4, max_contacts=None):
    """A prismatic base (vertical slide) carrying a chain of revolute links with position limits [-0.5, 0.4]; every joint enforces its
    limits when `enforce`.  With `ground`: a ground box under it and a box collider on the base, which touches the ground for slide positions
    in (-0.03, 0] (contacts and limit rows in one LCP)."""
    bodies = [na.BodySpec("base", -1, "prismatic", "slide", axis=(0, 1, 0), mass=1.0, inertia=(0.01, 0.01, 0.01, 0, 0, 0), pos_lo=(-0.2,), pos_hi=(0.6,),
                          limit_enforced=enforce)]
    for k in range(n_links):
        bodies.append(na.BodySpec(f"link{k}", k, "revolute", f"hinge{k}", axis=(0, 0, 1) if k % 2 == 0 else (1, 0, 0),
                                  T_pj=na.make_transform((0.2 if k else 0.0, 0, 0)), T_cj=na.make_transform((-0.1, 0, 0)), mass=0.5,
                                  inertia=(0.002, 0.003, 0.004, 0, 0, 0), damping=(0.05,), pos_lo=(-0.5,), pos_hi=(0.4,), limit_enforced=enforce))
    boxes = []
    if ground:
        boxes = [na.BoxSpec(-1, na.make_transform((0, -0.55, 0)), (6.0, 1.0, 6.0), 1.0),
                 na.BoxSpec(0, np.eye(4), (0.2, 0.1, 0.2), 0.8)]
    # (with the ground: four corner contacts + up to five limit rows - more than 8 constraint slots: the 48-row build)
    return na.ModelDescription("limited_arm", bodies, boxes, max_contacts=max_contacts or (16 if ground else 8))


# ---- capsule colliders (tests/test_oracle_capsules.py, tests/test_gpu_capsules.py) ----
def capsule_world(order="fixed_first", kinds=("capsule",), radius=0.1, height=0.4):
    """Free-joint capsules (axis = their z, radius 0.1, cylinder height 0.4) / spheres (radius 0.1) over a world-fixed capsule of radius
    0.25 whose axis is the world's z axis through the origin (a capsule cannot meet a box: that pair is libccd's in the reference)."""
    bodies, cols = [], []
    ground = na.CapsuleSpec(-1, np.eye(4), 0.25, 3.0, 1.0)
    for i, kind in enumerate(kinds):
        I = 0.4 * radius * radius
        bodies.append(na.BodySpec(f"{kind}{i}", -1, "free", f"{kind}{i}_joint", mass=1.0, inertia=(I, I, 0.5 * I, 0, 0, 0)))
        cols.append(na.CapsuleSpec(i, np.eye(4), radius, height, 0.8) if kind == "capsule" else na.SphereSpec(i, np.eye(4), radius, 0.8))
    boxes = [ground] + cols if order == "fixed_first" else cols + [ground]
    return na.ModelDescription("capsules", bodies, boxes, max_contacts=8)


def ball_state(md, centres, seed, pen=2e-3, radius=0.1):
    rng = np.random.default_rng(seed)
    n = md.num_dofs
    q = np.zeros(n); v = rng.normal(0, 0.02, n)
    for i, (x, z) in enumerate(centres):
        q[6 * i + 0:6 * i + 3] = rng.normal(0, 0.3, 3)
        q[6 * i + 3] = x; q[6 * i + 4] = radius - pen; q[6 * i + 5] = z
    return np.concatenate([q, v]), rng.normal(0, 0.1, n)




# ---- random frictional-contact LCPs for the solver tests (host emulation and GPU self-test) ----
def contact_lcp(rng, nc, ndof, cfm=0.0):
    """A = J D J^T (+ cfm I) of nc frictional contacts on an ndof-DOF system (rank-deficient when ndof < 3 nc), random b,
    bounds as DantzigBoxedLcpSolver::solve receives them (friction rows: lo = -mu, hi = mu, findex = their normal row)."""
    n = 3 * nc
    J = rng.normal(0, 1, (n, ndof))
    A = J @ np.diag(rng.uniform(0.1, 2, ndof)) @ J.T + cfm * np.eye(n)
    b = rng.normal(0, 1, n) * rng.choice([1, 0.01])
    mu = rng.choice([0.5, 1.0], nc)
    lo = np.zeros(n); hi = np.full(n, np.inf); fi = np.full(n, -1, np.int32)
    for c in range(nc):
        for k in (1, 2):
            lo[3 * c + k] = -mu[c]; hi[3 * c + k] = mu[c]; fi[3 * c + k] = 3 * c
    return np.ascontiguousarray(A, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64), lo, hi, fi


def have_ref():
    """oracle/_ref (the reference's own Dantzig solver compiled from its vendored sources) is present."""
    import os
    import oracle
    return os.path.exists(os.path.join(os.path.dirname(oracle.__file__), "_ref", "libodelcp_ref.so"))


def duplicate_filter_scene(deep=4):
    """One free body carrying `deep` + 1 small boxes in a row above a ground plate: the first `deep` sit 0.05 deep in the ground (their
    four bottom corners are narrow-phase points that the depth filter drops - 0.03 is the clipping depth), the last one 0.01 deep (four
    contacts).  With deep = 4 the kept contacts arrive as distinct points 17..20 of the world: past the 16 the device's duplicate
    filter remembers (NBL_ST_CONTACT_OVERFLOW); with deep = 3 as points 13..16: no flag.  -> (model, state [1, 12], action [1, 6])."""
    import nimblephysics_amd as na
    I = (0.02, 0.02, 0.02, 0.0, 0.0, 0.0)
    bodies = [na.BodySpec("carrier", -1, "free", "root", mass=1.0, inertia=I)]
    boxes = [na.BoxSpec(-1, na.make_transform((0.0, -0.5, 0.0)), (10.0, 1.0, 10.0), 1.0)]
    for k in range(deep + 1):
        down = 0.05 if k < deep else 0.01
        boxes.append(na.BoxSpec(0, na.make_transform((0.3 * k, 0.05 - down, 0.0)), (0.1, 0.1, 0.1), 1.0))
    md = na.ModelDescription("duplicate_filter", bodies, boxes, gravity=(0.0, -9.81, 0.0), dt=1e-3, max_contacts=8)
    s = np.zeros((1, 12)); s[0, 1] = 0.01                        # a little yaw: no two corners share a coordinate
    return md, s, np.zeros((1, 6))


# ---- more than 8 contacts per world: the 48-row instantiation of the library (tests/test_gpu_contacts16.py) ----
def cube_world(n_cubes=3, max_contacts=16, side=0.2, mass=0.1, mu=1.0):
    """A welded ground plate (top face y = 0) and `n_cubes` FreeJoint cubes (zero configuration: centre at the origin)."""
    I = mass * side * side / 6.0
    bodies = [na.BodySpec(f"cube{i}", -1, "free", f"cube{i}_joint", mass=mass, inertia=(I, I, I, 0, 0, 0)) for i in range(n_cubes)]
    boxes = [na.BoxSpec(-1, na.make_transform((0, -0.5, 0)), (6.0, 1.0, 6.0), mu)]
    boxes += [na.BoxSpec(i, np.eye(4), (side, side, side), mu) for i in range(n_cubes)]
    return na.ModelDescription("cubes", bodies, boxes, gravity=(0.0, -9.81, 0.0), dt=1e-3, max_contacts=max_contacts)


def cube_tower_inputs(B, seed, n_cubes=3, side=0.2, **kw):
    """A tower of cubes on the ground plate: 4 contacts per interface (FACE_VERTEX / VERTEX_FACE / EDGE_EDGE between cubes of one yaw and a
    small lateral offset), 12 contacts with three cubes, 16 with four."""
    rng = np.random.default_rng(seed)
    md = cube_world(n_cubes, side=side, **kw)
    n = 6 * n_cubes
    q = np.zeros((B, n)); v = np.zeros((B, n))
    yaw = rng.uniform(-1.0, 1.0, B)
    x, z = rng.uniform(-0.3, 0.3, B), rng.uniform(-0.3, 0.3, B)
    c, s_ = np.cos(yaw), np.sin(yaw)
    y = np.zeros(B)
    for k in range(n_cubes):
        y = y + (0.5 * side if k == 0 else side) - rng.uniform(1e-4, 1e-3, B)
        off = (rng.uniform(0.005, 0.03, (B, 2)) * rng.choice([-1, 1], (B, 2))) if k else np.zeros((B, 2))
        x = x + c * off[:, 0] + s_ * off[:, 1]; z = z - s_ * off[:, 0] + c * off[:, 1]
        q[:, 6 * k + 1] = yaw; q[:, 6 * k + 3] = x; q[:, 6 * k + 4] = y; q[:, 6 * k + 5] = z
        v[:, [6 * k + 3, 6 * k + 5]] = rng.normal(0, 0.05, (B, 2))
    return md, np.concatenate([q, v], 1), np.zeros((B, n))


def table_inputs(B, seed, feet=4, max_contacts=16):
    """One free body standing on `feet` small boxes (4 contacts each: 16 with four feet), randomly pushed: 48 LCP rows of rank 6."""
    rng = np.random.default_rng(seed)
    I = (0.05, 0.08, 0.05, 0.0, 0.0, 0.0)
    bodies = [na.BodySpec("table", -1, "free", "root", mass=2.0, inertia=I)]
    boxes = [na.BoxSpec(-1, na.make_transform((0.0, -0.5, 0.0)), (10.0, 1.0, 10.0), 1.0)]
    corners = [(-0.3, -0.2), (0.3, -0.2), (-0.3, 0.2), (0.3, 0.2), (0.0, 0.0)][:feet]
    for (cx, cz) in corners:
        boxes.append(na.BoxSpec(0, na.make_transform((cx, 0.05, cz)), (0.1, 0.1, 0.1), 1.0))
    md = na.ModelDescription("table", bodies, boxes, gravity=(0.0, -9.81, 0.0), dt=1e-3, max_contacts=max_contacts)
    s = np.zeros((B, 12))
    s[:, 1] = rng.uniform(-1.0, 1.0, B)                                   # yaw
    s[:, 3] = rng.uniform(-1, 1, B); s[:, 5] = rng.uniform(-1, 1, B)
    s[:, 4] = -rng.uniform(1e-4, 2e-3, B)                                 # feet 0.1 .. 2 mm in the ground
    s[:, 6:] = rng.normal(0, 0.05, (B, 6)) * np.where(rng.random((B, 1)) < 0.7, 1.0, 0.02)   # most pushed / spun, some nearly at rest
    return md, s, rng.normal(0, 0.2, (B, 6))
