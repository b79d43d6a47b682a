"""More than 8 contacts per world (the reference keeps every contact of every collider pair, ConstraintSolver.cpp:563-606; up to 8 per box
pair, DARTCollide.cpp:1384-1448): models with max_contacts up to 16, up to 32 colliders and 64 collider pairs run the 48-row instantiation
of the library's contact stage (csrc/abi_variants.h).  Every world against the oracle, which solves with all the contacts there are;
no world may carry NBL_ST_CONTACT_OVERFLOW and none is masked."""
import os

import numpy as np
import pytest

from parity import assert_match_or_reference_unstable, world_errors

pytestmark = pytest.mark.gpu
TOL = 1e-7
# The perturbation the reference-instability proofs may use, and how much closer to one of the oracle's perturbed outcomes than they scatter
# a device result has to be where those form a continuum.  Rounds 3-4 ran these files at 16 ulps / closeness 1.0; at 4 ulps / 0.25
#  nothing breaks: every file passes with the same world counts (profiles/r05_hatches_ulps4.log).
ULPS = int(os.environ.get("NBL_TEST_ULPS", "4"))
CLOSENESS = float(os.environ.get("NBL_TEST_CLOSENESS", "0.25"))


def _fwd_bwd(md, s, a, seed):
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    world = na.World(md, device="cuda:0")
    g = np.random.default_rng(seed).normal(0, 1, s.shape)
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy().astype(np.uint32)
    out.backward(torch.tensor(g, device="cuda:0"))
    dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
    return world, dev, status, g


def _contacts_of(world):
    return world._L.nbl_model_max_contacts(world._h)


def test_the_metric_model_on_the_48_row_build_equals_the_oracle_and_the_24_row_build():
    """Atlas-20 on the ground, metric distribution (half of the worlds through the fallback cascade), created with max_contacts = 16:
    the same worlds through the other instantiation of every dense kernel (48 x 49 LDS tiles, 64-bit row masks, 3 x 3 MFMA tiles, the
    two-pass Householder route, PGS sweeps of 16 / 32 / 48 rows)."""
    from oracle import OracleWorld
    from util import contact_inputs
    B = 512
    md, s, a = contact_inputs("atlas20", B, 31, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
    w8, dev8, st8, g = _fwd_bwd(md, s, a, 32)
    md.max_contacts = 16
    w16, dev16, st16, _ = _fwd_bwd(md, s, a, 32)
    assert _contacts_of(w8) == 8 and _contacts_of(w16) == 16
    ow = OracleWorld(md)
    ref = ow.step_batch(s, a, g, threads=8)
    assert np.array_equal(st16 & 0x83, ref["status"] & 0x83) and not (st16 & 0x80).any()
    assert 0.3 < ((st16 & 0x2) != 0).mean() < 0.7
    bad, _ = assert_match_or_reference_unstable("atlas20 sigma=0.02 on the 48-row build", ow, s, a, g, dev16, ref, TOL)
    assert bad <= 0.01 * B
    # the two builds: same stage for every world, stage-0 worlds to round-off (the arithmetic differs in order only)
    assert (st8 == st16).mean() > 0.99
    e, _ = world_errors(dev16, dev8)
    s0 = ((st8 & 0x2) != 0) & (st8 == st16)
    for k in e:
        assert e[k][s0].max() < 1e-9, (k, float(e[k][s0].max()))


@pytest.mark.parametrize("n_cubes,B,seed", [(3, 1024, 41), (4, 512, 42)])
def test_cube_towers_of_12_and_16_contacts(n_cubes, B, seed):
    """Three / four stacked cubes: 12 / 16 contacts in ONE constrained group, rank-deficient Delassus matrix of 36 / 48 rows."""
    from oracle import OracleWorld
    from util import cube_tower_inputs
    md, s, a = cube_tower_inputs(B, seed, n_cubes)
    world, dev, st, g = _fwd_bwd(md, s, a, seed + 1)
    assert _contacts_of(world) == 16
    ow = OracleWorld(md)
    ref = ow.step_batch(s, a, g, threads=8)
    assert not (st & 0x80).any() and not (ref["status"] & 0x80).any()
    assert np.array_equal(st & 0x1, ref["status"] & 0x1) and (st & 0x1).all()
    same = (st & 0x13e) == (ref["status"] & 0x13e)
    print(f"[{n_cubes} cubes] stage histogram (device):", {hex(int(k)): int(v) for k, v in zip(*np.unique(st & 0x13e, return_counts=True))}, "same stage as the oracle:", float(same.mean()))
    bad, _ = assert_match_or_reference_unstable(f"{n_cubes}-cube tower", ow, s, a, g, dev, ref, TOL, ulps=ULPS, n_perturb=64)
    assert bad <= 0.03 * B


def test_a_table_on_four_feet_has_sixteen_contacts_of_rank_six():
    """One free body on 16 coplanar contact points: A is 48 x 48 of rank 6 - the rank decisions of both pseudo-inverse routes at full size,
    and the most degenerate LCP this path can be given (16 normal impulses, 3 equations).  Worlds whose guess is a valid solution (stage 0,
    two thirds) must agree with the oracle to 1e-7 and be the same worlds.  In the others the REFERENCE's own answer flips between the
    pivoting stage, the CFM stage and the failed cascade under 16-ulp perturbations of the state (7 % of the batch: the Dantzig driver's
    early exit on a singular A(C,C) is decided by round-off, and a solved degenerate LCP has many solutions): those must be one of the
    reference's own outcomes - within 1e-7 of a perturbed run of the oracle, or no farther from one than those runs scatter."""
    from oracle import OracleWorld
    from util import table_inputs
    B = 512
    md, s, a = table_inputs(B, 51)
    world, dev, st, g = _fwd_bwd(md, s, a, 52)
    ow = OracleWorld(md)
    ref = ow.step_batch(s, a, g, threads=8)
    assert (st & 0x1).all() and not (st & 0x80).any() and not (ref["status"] & 0x80).any()
    stage0 = (st & 0x2) != 0
    assert np.array_equal(stage0, (ref["status"] & 0x2) != 0) and stage0.mean() > 0.5
    e, _ = world_errors(dev, ref)
    for k in e:
        assert e[k][stage0].max() < TOL, (k, float(e[k][stage0].max()))
    bad, by_closeness = assert_match_or_reference_unstable("table on 16 contacts", ow, s, a, g, dev, ref, TOL, ulps=ULPS, closeness=CLOSENESS, max_by_closeness=int(0.02 * B))
    # 33 of 512 worlds (6.4 %) on this seed, each one PROVEN above (the oracle's own answer moves by more than 1e-7 under 4-ulp perturbations
    # of that world's state, and the device result is within 1e-7 of one of those outcomes: 31, or 4 x closer to one than they scatter: 2).
    # Two-sided, so that a change of the cascade in either direction shows up
    assert 0.03 * B <= bad <= 0.08 * B and by_closeness <= 0.02 * B, (bad, by_closeness)


def test_more_than_sixteen_colliders_run_the_48_row_build_too():
    """20 colliders (the reference's biped.skel has 20, fullbody1.skel 21) with at most 8 contacts: the collider table, not the row count,
    sends this model to the larger instantiation."""
    import nimblephysics_amd as na
    from oracle import OracleWorld
    rng = np.random.default_rng(61)
    I = (0.02, 0.02, 0.02, 0.0, 0.0, 0.0)
    bodies = [na.BodySpec("carrier", -1, "free", "root", mass=1.0, inertia=I)]
    boxes = [na.BoxSpec(-1, na.make_transform((0.0, -0.5, 0.0)), (10.0, 1.0, 10.0), 1.0)]
    for k in range(19):          # two of them low enough to touch the ground, the others above it
        low = k in (3, 11)
        boxes.append(na.BoxSpec(0, na.make_transform((0.25 * (k % 5), 0.05 + (0.0 if low else 0.2), 0.25 * (k // 5))), (0.1, 0.1, 0.1), 0.9))
    md = na.ModelDescription("twenty", bodies, boxes, gravity=(0.0, -9.81, 0.0), dt=1e-3, max_contacts=8)
    B = 256
    s = np.zeros((B, 12)); s[:, 1] = rng.uniform(-1, 1, B); s[:, 4] = -rng.uniform(1e-4, 2e-3, B); s[:, 6:] = rng.normal(0, 0.05, (B, 6))
    a = rng.normal(0, 0.2, (B, 6))
    world, dev, st, g = _fwd_bwd(md, s, a, 62)
    assert _contacts_of(world) == 16
    ow = OracleWorld(md)
    ref = ow.step_batch(s, a, g, threads=8)
    assert (st & 0x1).all() and np.array_equal(st & 0x81, ref["status"] & 0x81)
    bad, _ = assert_match_or_reference_unstable("20 colliders", ow, s, a, g, dev, ref, TOL, ulps=ULPS)
    assert bad <= 0.03 * B
