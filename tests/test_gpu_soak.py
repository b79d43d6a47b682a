"""Randomised parity soak (tools/soak_parity.py): random trees, random colliders (boxes / spheres; friction from frictionless to
1.5; restitution; penetration correction on / off) dropped on the ground at random heights and speeds.  EVERY world is compared
with the oracle (next state and both gradients); a world above 1e-5 must be one where the oracle itself flips under 1-ulp input
perturbations and the device equals one of its outcomes.  (The full soak of the round: 600 models x 512 worlds = 307 200 worlds,
64 798 in contact, 47 716 through the fallback cascade: 11 above 1e-7, 10 above 1e-5 - all 10 reference-unstable, 0 mismatches.)"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
# Worlds that pass the proof-of-instability only by being 10 x closer to one of the oracle's perturbed outcomes than those scatter (not
# within tol of one): counted by tools/soak_parity.py and bounded here ("by_closeness is printed but never bounded")
BY_CLOSENESS_MAX = 2
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_models_all_worlds_vs_oracle():
    import soak_parity
    tot = soak_parity.run(1000, 100, 256, verbose=False)
    print(tot)
    assert tot["MISMATCH"] == 0 and tot["by_closeness"] <= BY_CLOSENESS_MAX, tot
    assert tot["contact"] > 0.1 * tot["worlds"] and tot["cascade"] > 0.05 * tot["worlds"], tot       # the soak does reach the cascade
    assert tot["gt1e-7"] <= 0.001 * tot["worlds"], tot


def test_random_larger_models_with_many_colliders_all_worlds_vs_oracle():
    """8-21 bodies, 3-7 colliders, 16 contact slots per world (the 48-row build of the library; with 8 slots 0.2 % of these worlds held a
    ninth contact, were truncated, flagged and left out of the comparison in round 3): no world overflows, none is left out.
    Soak of round 3 (8 slots): 300 models x 256 worlds = 76 800 worlds, 29 610 in contact, 21 434 through the cascade: 3 above 1e-5, all
    reference-unstable, 0 mismatches."""
    import soak_parity
    tot = soak_parity.run(7000, 40, 256, verbose=False, big=True)
    print(tot)
    assert tot["MISMATCH"] == 0 and tot["overflow"] == 0 and tot["by_closeness"] <= BY_CLOSENESS_MAX, tot
    assert tot["contact"] > 0.2 * tot["worlds"], tot


def test_random_worlds_with_several_skeletons_all_worlds_vs_oracle():
    """Two or three separate skeletons per world, each a small random tree with its own colliders: several constrained groups
    (ConstraintSolver.cpp:724-780), every one with its own run of the solver cascade.  Soak of the round: 150 models x 256 worlds =
    38 400 worlds, 16 568 in contact, 12 229 through the cascade: 6 above 1e-5, all reference-unstable, 0 mismatches."""
    import soak_parity
    tot = soak_parity.run(9500, 40, 256, verbose=False, multi=True)
    print(tot)
    assert tot["MISMATCH"] == 0 and tot["by_closeness"] <= BY_CLOSENESS_MAX, tot
    assert tot["contact"] > 0.2 * tot["worlds"] and tot["cascade"] > 0.1 * tot["worlds"], tot


def test_random_models_with_ball_joints_all_worlds_vs_oracle():
    """40 % of the joints below the free root are ball joints (three device bodies each + the closed-form acceleration term; the oracle
    runs the reference's 3-DOF joint), colliders on random bodies."""
    import soak_parity
    tot = soak_parity.run(9000, 40, 256, verbose=False, balls=True)
    print(tot)
    assert tot["MISMATCH"] == 0 and tot["by_closeness"] <= BY_CLOSENESS_MAX, tot
    assert tot["contact"] > 0.1 * tot["worlds"], tot
    assert tot["gt1e-7"] <= 0.002 * tot["worlds"], tot


def test_random_models_ten_metres_from_the_world_origin_all_worlds_vs_oracle():
    """The same scenes (with ball joints) in a corner of the ground plate, ~10 m from the world origin: the rank decisions of the contact
    solver must not depend on where the scene sits (spatial quantities are carried about the root of each tree).  Soak of the round: 600
    models x 256 worlds = 153 600 worlds, 31 926 in contact: 8 above 1e-7, 7 of them reference-unstable, 0 mismatches."""
    import soak_parity
    tot = soak_parity.run(11000, 30, 256, verbose=False, balls=True, far=True)
    print(tot)
    assert tot["MISMATCH"] == 0 and tot["by_closeness"] <= BY_CLOSENESS_MAX, tot
    assert tot["contact"] > 0.1 * tot["worlds"], tot
    assert tot["gt1e-7"] <= 0.002 * tot["worlds"], tot


def test_dense_step_jacobians_of_random_models_vs_oracle():
    """tools/soak_jacobians.py: every column of d next / d state and d next / d action (2n vector-Jacobian products through the snapshot
    interface) against the oracle's World::getStateJacobian / getActionJacobian on random models with ball and free joints, in and out of
    contact.  Soak of the round: 150 models x 4 worlds: worst 1.9e-9."""
    import soak_jacobians
    tot = soak_jacobians.run(12000, 25, 4, "balls", verbose=False)
    print(tot)
    assert tot["worlds"] >= 80 and tot["contact"] > 0
    assert tot["gt1e-5"] == 0 and tot["gt1e-7"] <= 1, tot


def test_warm_started_second_step_of_random_models_vs_oracle():
    """tools/soak_warm.py: a cold step, then a second step whose LCP starts from the first step's solution (the reference's solver carries
    mX between steps), device and oracle each with their own; every world of the second step against the oracle.  Soak of the round: 600
    models (balls + multi) x 256: 0 mismatches."""
    import soak_warm
    tot = soak_warm.run(13000, 30, 256, "balls", verbose=False)
    print(tot)
    assert tot["MISMATCH"] == 0, tot
    assert tot["contact2"] > 0.1 * tot["worlds"] and tot["stage0"] > 0.3 * tot["contact2"], tot      # the warm start does resolve worlds at stage 0


def test_the_references_finite_difference_noise_away_from_pi_is_proven_by_exact_derivatives():
    """Round 6's only soak hit (warm:mix, seed 545103, world 6 - a free-floating tree with NO contact): next state and the action gradient
    equal to round-off, ONE entry of the state gradient (the root's angular velocity about z) 3.4e-4 away from the oracle.  The reference
    differentiates the position integration of a free joint by central differences with eps = 1e-6 (FreeJoint.cpp:950-1007; the oracle
    restates it) of a logMap that takes the angle of the step's rotation increment (|w| dt ~ 1e-3 rad) from an arc cosine: 1e-10 of noise
    in the forward pass - identical bits on both sides - amplified by 1 / eps.  Asserted here: the device's gradient (a) differs from the
    finite-difference oracle by more than 1e-5 in that world, (b) equals the oracle WITH its exact-derivative instrument to 1e-7, (c)
    equals central differences of the device's own forward step at h = 1e-4 to 1e-5; and the soak files such a world under
    `reference_fd_exact_agrees`, not MISMATCH (tools/soak_parity.py::exact_derivatives_agree)."""
    import numpy as np
    import torch
    import nimblephysics_amd as na
    import soak_parity
    import soak_stress
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    seed, wd, B = 545103, 6, 256
    md, s, a, g = soak_parity.make_case(seed, B, balls=True)
    md, s, a, g = soak_stress.mutator("mix")(seed, md, s, a, g)
    world = na.World(md, device="cuda:0")
    ow = OracleWorld(md)
    at = torch.tensor(a, device="cuda:0")
    with torch.no_grad():
        s1 = timestep(world, torch.tensor(s, device="cuda:0"), at)
    assert not (world.last_status.cpu().numpy()[wd] & 1), "no contact in this world"
    world.reset_lcp_cache()
    st = s1.clone().requires_grad_(True); at2 = at.clone().requires_grad_(True)
    out = timestep(world, st, at2)
    out.backward(torch.tensor(g, device="cuda:0"))
    s1n = s1.cpu().numpy()
    dev = {"next": out.detach().cpu().numpy()[wd], "grad_state": st.grad.cpu().numpy()[wd], "grad_action": at2.grad.cpu().numpy()[wd]}
    fd = ow.step_batch(s1n[wd:wd + 1], a[wd:wd + 1], g[wd:wd + 1])
    ow.set_exact_position_jacobians(True)
    ex = ow.step_batch(s1n[wd:wd + 1], a[wd:wd + 1], g[wd:wd + 1])
    ow.set_exact_position_jacobians(False)
    sc = {k: max(np.abs(fd[k]).max(), 1e-30) for k in dev}
    e_fd = {k: float(np.abs(dev[k] - fd[k][0]).max() / sc[k]) for k in dev}
    e_ex = {k: float(np.abs(dev[k] - ex[k][0]).max() / sc[k]) for k in dev}
    print("[finite-difference noise away from pi] device vs the reference's differences:", e_fd, " vs exact derivatives:", e_ex)
    assert e_fd["next"] < 1e-12 and e_fd["grad_action"] < 1e-12 and e_fd["grad_state"] > 1e-5
    assert max(e_ex.values()) < 1e-7
    e = int(np.abs(dev["grad_state"] - fd["grad_state"][0]).argmax())
    h = 1e-4
    sp = s1n.copy(); sm = s1n.copy(); sp[wd, e] += h; sm[wd, e] -= h
    with torch.no_grad():
        world.reset_lcp_cache(); fp = timestep(world, torch.tensor(sp, device="cuda:0"), at).cpu().numpy()[wd]
        world.reset_lcp_cache(); fm = timestep(world, torch.tensor(sm, device="cuda:0"), at).cpu().numpy()[wd]
    num = float(np.dot(g[wd], (fp - fm) / (2 * h)))
    assert abs(num - dev["grad_state"][e]) < 1e-5 * sc["grad_state"], (num, dev["grad_state"][e], fd["grad_state"][0][e])
    assert soak_parity.exact_derivatives_agree(ow, 1e-6, s1n[wd], a[wd], g[wd], dev, sc)
