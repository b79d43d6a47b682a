"""Stress soaks at reduced size (tools/soak_stress.py and the former tools/dbg one-off drivers): random models at the edges of the
parameter space, z-up scenes, exponential-map joints exactly at zero, the largest models the contact path accepts, mass gradients of
random models, create / destroy cycles.  Every world against the oracle; the full-size runs are the tools' `__main__`."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _fwd_bwd_vs_oracle(md, s, a, g):
    """-> (per-world max relative error over next state and both gradients with contact-overflow worlds zeroed, device status)."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy().astype(np.uint32)
    out.backward(torch.tensor(g, device="cuda:0"))
    ref = OracleWorld(md).step_batch(s, a, g, threads=8)
    dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
    assert all(np.isfinite(x).all() for x in dev.values())
    err = np.maximum.reduce([np.abs(dev[k] - ref[k]).max(1) / max(np.abs(ref[k]).max(), 1e-30) for k in dev])
    err[((status | ref["status"]) & 0x80) != 0] = 0
    return err, status


@pytest.mark.parametrize("mode", ["dt", "tinydt", "fast", "torque", "mass", "nograv", "geom", "mu", "subset", "atlimit", "capsule", "limits", "selfcol", "adjacent", "mix"])
def test_stress_variants_of_the_random_soak_all_worlds_vs_oracle(mode):
    """(Round 2 at full size, 150 models x 256 worlds per mode: 368 640 worlds, 0 mismatches.)"""
    import soak_stress
    tot = soak_stress.run(mode, 20000, 8, 128)
    print(mode, tot)
    assert tot["MISMATCH"] == 0, tot
    assert tot["worlds"] >= 512 and tot["contact"] > 0, tot
    assert tot["gt1e-5"] <= 0.005 * tot["worlds"], tot


def test_a_world_that_exhausts_the_duplicate_filters_memory_is_flagged():
    """The reference's postProcess drops a contact point that coincides with ANY point its detector has produced so far, also with the
    ones the depth filter removes later (DARTCollisionDetector.cpp:360-400); the device remembers 2 x its contact slots distinct points per
    world (model_dev.hpp SEEN_POINTS: 16 on the 24-row build this test pins the model to, 32 on the 48-row build).  A folded 21-body tree with big colliders deep in one another produces 25: a contact kept after the
    list is full may be an unnoticed duplicate (here: two vertices of one box that touch two other boxes), so the world carries
    NBL_ST_CONTACT_OVERFLOW - on the device and, by the same rule, in the oracle (round 3: found by the mixed-feature soak as a world
    whose two extra contacts went unflagged).  Every unflagged world of the batch agrees with the oracle."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    import soak_parity
    import soak_stress
    mode = "geom+mass+selfcol+limits+dt"                              # (what the mixed mode drew for this seed when it found the world)
    tot = soak_stress.run(mode, 160020, 1, 256, variant="big", slots=8)        # (asserts the overflow flags world by world)
    assert tot["MISMATCH"] == 0, tot
    md, s, a, g = soak_parity.make_case(160020, 256, True, False, False, False, slots=8)
    md, s, a, g = soak_stress.mutator(mode, slots=8)(160020, md, s, a, g)
    world = na.World(md, device="cuda:0")
    timestep(world, torch.tensor(s, device="cuda:0"), torch.tensor(a, device="cuda:0"))
    status = world.last_status.cpu().numpy().astype(np.uint32)
    ref = OracleWorld(md).step_batch(s, a, None, threads=8)
    assert (status[152] & 0x80) and (ref["status"][152] & 0x80)
    assert not (ref["status"][152] & 0x1) and (status[152] & 0x1)      # the reference has no contact there; the device keeps the two duplicates


def test_z_up_scenes_take_the_fallback_branch_of_the_tangent_basis():
    """Gravity along -z, ground normal +-z: ContactConstraint::getTangentBasisMatrixODE (ContactConstraint.cpp:734-795) takes its
    fallback branch (normal parallel to the first candidate axis)."""
    import nimblephysics_amd as na
    from test_gpu_random_trees import random_tree
    contact = bad = 0
    for seed in range(10):
        rng = np.random.default_rng(7000 + seed)
        md = random_tree(rng, int(rng.integers(1, 6)), "random", True, colliders=int(rng.integers(1, 4)), spheres=bool(rng.random() < 0.5), balls=0.3)
        md.gravity = (0.0, 0.0, -9.81)
        md.boxes[0] = na.BoxSpec(-1, na.make_transform((0, 0, -0.005)), (20.0, 20.0, 0.01), 1.0)
        B = 128; n = md.num_dofs
        q = rng.normal(0, 0.25, (B, n)); q[:, 3] = rng.normal(0, 0.3, B); q[:, 4] = rng.normal(0, 0.3, B); q[:, 5] = rng.uniform(0.02, 0.4, B)
        s = np.concatenate([q, rng.normal(0, 0.5, (B, n))], 1); a = rng.normal(0, 0.5, (B, len(md.action_map))); g = rng.normal(0, 1, s.shape)
        err, status = _fwd_bwd_vs_oracle(md, s, a, g)
        contact += int((status & 1).sum()); bad += int((err > 1e-5).sum())
    print("z-up: worlds in contact", contact, "above 1e-5:", bad)
    assert contact > 100 and bad == 0


def test_exponential_map_joints_exactly_at_zero_and_at_the_taylor_branch_boundary():
    """Ball joints at q = 0, w = 0, both, |q| = 1e-9 and |q| just below / above the 1e-3 switch of the Taylor branches of expMapRot /
    expMapJac / logMap (Geometry.cpp:539-600, 720-760) and of their reverse mode."""
    from test_ball_joint import _ball_offsets, ball_model
    md = ball_model(11, True); n = md.num_dofs; B = 64
    offs = [0] + _ball_offsets(md)
    for name, zq, zw, tiny in (("q = 0", True, False, 0), ("w = 0", False, True, 0), ("q = w = 0", True, True, 0), ("|q| = 1e-9", True, False, 1e-9),
                               ("|q| = 9.99e-4 / 1.001e-3", True, False, 1e-3)):
        rng = np.random.default_rng(5)
        q = rng.normal(0, 0.5, (B, n)); v = rng.normal(0, 1.0, (B, n))
        for o in offs:
            if zq:
                ax = rng.normal(size=(B, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
                q[:, o:o + 3] = ax * (tiny * (1 + 2e-3 * (np.arange(B)[:, None] % 2 - 0.5)) if tiny else 0.0)
            if zw:
                v[:, o:o + 3] = 0.0
        s = np.concatenate([q, v], 1); a = rng.normal(0, 1, (B, len(md.action_map))); g = rng.normal(0, 1, s.shape)
        err, _ = _fwd_bwd_vs_oracle(md, s, a, g)
        print(f"{name:28s} max err {err.max():.1e}")
        assert err.max() < 1e-7, (name, err.max())


@pytest.mark.parametrize("shape", ["chain", "star", "random"])
def test_the_largest_models_of_the_contact_path_39_dofs_34_device_bodies(shape):
    """Eleven ball joints on a free root (39 DOFs, 1 + 3 * 11 = 34 device bodies) with five colliders: next to the 40-DOF / 64-body limits
    of the contact path."""
    import nimblephysics_amd as na
    from test_gpu_random_trees import random_tree
    rng = np.random.default_rng(1)
    md = random_tree(rng, 12, shape, True, colliders=5, spheres=True, balls=1.0)
    for b in md.bodies[1:]:
        b.joint_type = "ball"; b.damping = (); b.spring = (); b.rest = ()
    md = na.ModelDescription("limit", md.bodies, md.boxes, gravity=md.gravity, dt=md.dt, max_contacts=8)
    n = md.num_dofs; B = 128
    assert n == 39
    q = rng.normal(0, 0.3, (B, n)); q[:, 3] = rng.normal(0, 0.3, B); q[:, 5] = rng.normal(0, 0.3, B); q[:, 4] = rng.uniform(0.02, 0.5, B)
    s = np.concatenate([q, rng.normal(0, 0.5, (B, n))], 1); a = rng.normal(0, 0.5, (B, len(md.action_map))); g = rng.normal(0, 1, s.shape)
    err, status = _fwd_bwd_vs_oracle(md, s, a, g)
    print(shape, "in contact", (status & 1).mean(), "overflow", ((status & 0x80) != 0).mean(), "max err", err.max())
    assert (status & 1).mean() > 0.05 and err.max() < 1e-5


def test_models_up_to_the_64_dof_64_body_limits_of_the_contact_path():
    """lane = DOF / lane = body in the wavefront kernels: 64 of each (round 2 stopped at 40 DOFs).  A free root with 19 ball joints: 63 DOFs
    on 1 + 3 * 19 = 58 device bodies, four colliders; a chain of 57 revolute joints on a free root: 63 DOFs, 58 bodies."""
    import nimblephysics_amd as na
    from test_gpu_random_trees import random_tree
    for kind in ("balls", "revolutes"):
        rng = np.random.default_rng(2)
        md = random_tree(rng, 20 if kind == "balls" else 58, "random" if kind == "balls" else "chain", True, colliders=4, spheres=True, balls=1.0 if kind == "balls" else 0.0)
        if kind == "balls":
            for b in md.bodies[1:]:
                b.joint_type = "ball"; b.damping = (); b.spring = (); b.rest = ()
        md = na.ModelDescription("limit64", md.bodies, md.boxes, gravity=md.gravity, dt=md.dt, max_contacts=8)
        n = md.num_dofs; B = 64
        assert n == 63, n
        q = rng.normal(0, 0.2, (B, n)); q[:, 3] = rng.normal(0, 0.3, B); q[:, 5] = rng.normal(0, 0.3, B); q[:, 4] = rng.uniform(0.02, 0.5, B)
        s = np.concatenate([q, rng.normal(0, 0.3, (B, n))], 1); a = rng.normal(0, 0.3, (B, len(md.action_map))); g = rng.normal(0, 1, s.shape)
        err, status = _fwd_bwd_vs_oracle(md, s, a, g)
        print(kind, "n", n, "in contact", (status & 1).mean(), "overflow", ((status & 0x80) != 0).mean(), "max err", err.max())
        assert (status & 1).mean() > 0.05 and err.max() < 1e-5


def test_mass_gradients_of_random_models_vs_central_differences_of_the_oracle():
    """dL/dmass of random models (ball / free / revolute / prismatic joints, welds) in free fall, random bodies and entry types, against
    central differences of the oracle step with respect to the same parameters."""
    import soak_parity
    from nimblephysics_amd.mass import WrtMassBodyNodeEntryType as T
    from test_gpu_mass import _check
    done = 0
    for seed in range(6):
        case = soak_parity.make_case(seed, 16, balls=True)
        if case is None:
            continue
        md, s, a, g = case
        md.boxes = []; md.max_contacts = 0
        rng = np.random.default_rng(seed)
        movable = [i for i, b in enumerate(md.bodies) if b.joint_type != "weld"]
        entries = [(int(rng.choice(movable)), T(int(rng.choice([0, 1, 3, 4, 5])))) for _ in range(3)]
        entries = list({e[0]: e for e in entries}.values())
        _check(md, entries, s, a, seed + 1, tol=2e-5)
        done += 1
    assert done >= 4
    # ... and of the soak's mixed-feature models without their colliders and enforced limits (masses x 1e-2 .. 1e2, 5 ms steps, action
    # subsets, no gravity: the parameter extremes; fast / torque left out - central differences of a step with |v| ~ 20 carry no digits)
    import soak_stress
    done = 0
    for seed in range(600, 612):
        case = soak_parity.make_case(seed, 16, balls=True)
        if case is None:
            continue
        md, s, a, g = soak_stress.mutator("mass+subset+dt" if seed % 2 else "mass+nograv")(seed, *case)
        md.boxes = []; md.max_contacts = 0
        rng = np.random.default_rng(seed)
        movable = [i for i, b in enumerate(md.bodies) if b.joint_type != "weld"]
        entries = list({e[0]: e for e in [(int(rng.choice(movable)), T(int(rng.choice([0, 1, 3, 4, 5])))) for _ in range(3)]}.values())
        _check(md, entries, s, a, seed + 1, tol=5e-5, fd_relative=True)
        done += 1
    assert done >= 8


def test_creating_and_destroying_worlds_leaks_no_device_memory_and_changes_no_result():
    import gc
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from test_ball_joint import ball_model
    from util import contact_inputs
    torch.cuda.init()
    md, s, a = contact_inputs("atlas20", 256, 1)
    mb = ball_model(1, True, ground=True)
    ref = None
    used = []
    for it in range(40):
        m = md if it % 2 == 0 else mb
        w = na.World(m, device="cuda:0")
        n = m.num_dofs
        x = torch.tensor(s if it % 2 == 0 else np.random.default_rng(it).normal(0, 0.3, (64, 2 * n)), device="cuda:0", requires_grad=True)
        u = torch.tensor(a if it % 2 == 0 else np.zeros((64, len(m.action_map))), device="cuda:0")
        y = timestep(w, x, u); y.sum().backward()
        if it == 0:
            ref = y.detach().clone()
        if it % 2 == 0:
            assert torch.equal(ref, y.detach())
        w2 = w.clone(); del w, w2, x, y
        if it % 10 == 9:
            gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
            free, total = torch.cuda.mem_get_info()
            used.append((total - free) // 2 ** 20)
    print("device memory in use after every 10 create / destroy cycles (MB):", used)
    assert used[-1] <= used[0] + 64
