"""Random articulated trees (topology, joint types, frames, inertias, springs, dampers, limits, partial action spaces; with and
without box colliders on a ground box) through the C ABI against the CPU oracle.  The named configs exercise one topology
each; this covers what the level-synchronous lane = body kernels depend on: many siblings under one parent (LDS atomics on one
address), deep chains, several roots, free and fixed bases, welded links, padding of the body count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-7


def _rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _T(rng, scale):
    T = np.eye(4); T[:3, :3] = _rot(rng); T[:3, 3] = rng.normal(0, scale, 3)
    return T


def random_tree(rng, nb, shape, free_root, welds=0, colliders=0, spheres=False, balls=0.0):
    import nimblephysics_amd as na
    bodies = []
    for i in range(nb):
        if i == 0:
            parent = -1
        elif shape == "chain":
            parent = i - 1
        elif shape == "star":
            parent = 0
        else:
            parent = int(rng.integers(0, i))
        jt = "free" if (i == 0 and free_root) else ("prismatic" if rng.random() < 0.25 else "revolute")
        if welds and i > 0 and rng.random() < welds:
            jt = "weld"
        elif balls and i > 0 and rng.random() < balls:
            jt = "ball" if rng.random() < 0.8 else "free"          # ... and free joints below the root
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        nd = {"free": 6, "weld": 0, "ball": 3}.get(jt, 1)
        A = rng.normal(size=(3, 3)); I = A @ A.T * 0.05 + 0.05 * np.eye(3)
        bodies.append(na.BodySpec(
            f"b{i}", parent, jt, f"j{i}", axis=tuple(axis),
            T_pj=np.eye(4) if (i == 0 and free_root) else _T(rng, 0.3), T_cj=np.eye(4) if (i == 0 and free_root) else _T(rng, 0.1),
            mass=float(rng.uniform(0.5, 3.0)), com=tuple(rng.normal(0, 0.05, 3)),
            inertia=(I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]),
            damping=tuple(rng.uniform(0, 2, nd)) if rng.random() < 0.5 else (), spring=tuple(rng.uniform(0, 5, nd)) if rng.random() < 0.3 else (),
            rest=tuple(rng.normal(0, 0.1, nd)) if rng.random() < 0.3 else ()))
    boxes = []
    if colliders:
        boxes.append(na.BoxSpec(-1, na.make_transform((0, -0.005, 0)), (20.0, 0.01, 20.0), 1.0))
        movable = [i for i, b in enumerate(bodies)]
        for j, i in enumerate(rng.choice(movable, size=min(colliders, len(movable)), replace=False)):
            if spheres and j % 2 == 1:
                boxes.append(na.SphereSpec(int(i), na.make_transform(tuple(rng.normal(0, 0.03, 3))), float(rng.uniform(0.08, 0.2)), float(rng.uniform(0.5, 1.0))))
            else:
                boxes.append(na.BoxSpec(int(i), na.make_transform((0, 0, 0)), tuple(rng.uniform(0.1, 0.3, 3)), float(rng.uniform(0.5, 1.0))))
    md = na.ModelDescription("random_tree", bodies, boxes, gravity=(0.0, -9.81, 0.0), dt=1e-3, max_contacts=8 if colliders else 0)
    return md


def _compare(md, B, seed, action_dofs=None, tol=TOL):
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    rng = np.random.default_rng(seed)
    n = md.num_dofs
    if action_dofs is not None:
        md.set_action_space(action_dofs)
    k = len(md.action_map)
    s = np.concatenate([rng.normal(0, 0.4, (B, n)), rng.normal(0, 0.5, (B, n))], 1)
    a = rng.normal(0, 1.0, (B, k))
    g = rng.normal(0, 1, s.shape)
    world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    out.backward(torch.tensor(g, device="cuda:0"))
    ref = ow.step_batch(s, a, g, threads=8)
    sc = lambda x: max(np.abs(x).max(), 1e-30)
    errs = {"next": np.abs(out.detach().cpu().numpy() - ref["next"]).max() / sc(ref["next"]),
            "grad_state": np.abs(st.grad.cpu().numpy() - ref["grad_state"]).max() / sc(ref["grad_state"]),
            "grad_action": np.abs(at.grad.cpu().numpy() - ref["grad_action"]).max() / sc(ref["grad_action"])}
    assert max(errs.values()) < tol, (errs, n, k)
    return world, ref


@pytest.mark.parametrize("seed,nb,shape,free_root", [
    (1, 1, "chain", False), (2, 2, "chain", True), (3, 7, "chain", False), (4, 12, "star", False), (5, 13, "star", True),
    (6, 17, "random", True), (7, 24, "random", False), (8, 33, "random", True), (9, 5, "random", False), (10, 16, "chain", True),
])
def test_random_trees_without_contact(seed, nb, shape, free_root):
    rng = np.random.default_rng(1000 + seed)
    md = random_tree(rng, nb, shape, free_root)
    _compare(md, 96, seed)


@pytest.mark.parametrize("seed,nb,shape", [(21, 9, "random"), (22, 14, "star"), (23, 20, "random")])
def test_random_trees_with_welds_and_partial_action_space(seed, nb, shape):
    rng = np.random.default_rng(2000 + seed)
    md = random_tree(rng, nb, shape, free_root=bool(seed % 2), welds=0.3)
    n = md.num_dofs
    dofs = sorted(rng.choice(n, size=max(1, n // 2), replace=False).tolist())
    _compare(md, 64, seed, action_dofs=dofs)


@pytest.mark.parametrize("seed,nb,shape,spheres", [(31, 4, "chain", False), (32, 8, "random", False), (33, 11, "star", False),
                                                    (34, 6, "random", True), (35, 10, "chain", True)])
def test_random_trees_resting_on_the_ground(seed, nb, shape, spheres):
    """A free-root tree dropped so that some of its box colliders penetrate the ground box slightly: whatever contact set comes
    out (0..8 contacts, any mix of vertex / edge types), device and oracle must agree on state and gradients of the worlds
    whose LCP both resolved in stage 0 (the cascade's tie-breaks are compared in test_gpu_contact.py)."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    rng = np.random.default_rng(3000 + seed)
    md = random_tree(rng, nb, shape, free_root=True, colliders=4 if spheres else 3, spheres=spheres)
    n = md.num_dofs
    B = 128
    q = rng.normal(0, 0.2, (B, n)); q[:, 3] = rng.normal(0, 0.3, B); q[:, 5] = rng.normal(0, 0.3, B)
    q[:, 4] = rng.uniform(0.05, 0.6, B)      # height of the root: some worlds touch, some do not
    v = rng.normal(0, 0.05, (B, n))
    s = np.concatenate([q, v], 1); a = rng.normal(0, 0.2, (B, n)); g = rng.normal(0, 1, s.shape)
    world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy().astype(np.uint32)
    out.backward(torch.tensor(g, device="cuda:0"))
    ref = ow.step_batch(s, a, g, threads=8)
    assert np.array_equal(status & 0x1, ref["status"] & 0x1)            # same worlds in contact
    assert np.array_equal(status & 0x80, ref["status"] & 0x80)          # same contact overflow flags
    ok = ((status & 0x1) == 0) | (((status & 0x2) != 0) & ((ref["status"] & 0x2) != 0))
    ok &= (status & 0x80) == 0
    assert ok.mean() > 0.5
    assert ((status & 0x1) != 0).any()
    sc = lambda x: max(np.abs(x[ok]).max(), 1e-30)
    for name, dev, r in (("next", out.detach().cpu().numpy(), ref["next"]), ("grad_state", st.grad.cpu().numpy(), ref["grad_state"]),
                         ("grad_action", at.grad.cpu().numpy(), ref["grad_action"])):
        err = np.abs(dev[ok] - r[ok]).max() / sc(r)
        assert err < 1e-6, (name, err)
