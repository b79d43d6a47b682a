"""Joint-limit constraint rows in the CPU oracle (JointLimitConstraint.cpp:182-381, ConstraintSolver.cpp:641-696; opt-in through
BodySpec.limit_enforced = Joint::setPositionLimitEnforced): the forward LCP rows, and the reference's backward pass, in which a
joint-limit row has no constraint-force column and drops out of every Jacobian (DifferentiableContactConstraint.cpp:51-99)."""
import copy

import numpy as np
import pytest

import nimblephysics_amd as na
from oracle import OracleWorld
from util import limited_arm


def test_a_dof_at_its_limit_moving_outwards_is_stopped_and_the_impulse_reaches_the_other_dofs():
    md = copy.deepcopy(na.cartpole())
    md.bodies[1].pos_lo, md.bodies[1].pos_hi, md.bodies[1].limit_enforced = (-0.5,), (0.5,), True
    w = OracleWorld(md)
    free = copy.deepcopy(md); free.bodies[1].limit_enforced = False
    wf = OracleWorld(free)
    for q1, v1 in ((0.5, 1.0), (-0.5, -1.0), (0.6, 0.7), (-0.7, -0.2)):          # at / beyond the upper / lower limit, moving outwards
        s = np.array([0.0, q1, 0.3, v1]); a = np.zeros(w.k)
        nxt, nf = w.step(s, a), wf.step(s, a)
        assert w.last_status & 0x400 and not (w.last_status & 0x1)               # NBL_ST_JOINT_LIMIT, no contact
        assert abs(nxt[3]) < 1e-12 and abs(nf[3]) > 0.1                          # the pole's velocity is taken out ...
        assert abs(nxt[2] - nf[2]) > 1e-3                                        # ... and the sled feels the impulse
        assert np.array_equal(nxt[:2], nf[:2])                                   # positions integrate with the initial velocities
    for q1, v1 in ((0.5, -1.0), (-0.5, 1.0)):                                   # at the limit, moving inwards: the row is there, its impulse 0
        s = np.array([0.0, q1, 0.3, v1]); a = np.zeros(w.k)
        assert np.allclose(w.step(s, a), wf.step(s, a), rtol=0, atol=1e-14) and w.last_status & 0x400
    s = np.array([0.0, 0.2, 0.3, 1.0])                                          # inside the limits: no row
    assert np.array_equal(w.step(s, np.zeros(w.k)), wf.step(s, np.zeros(w.k))) and not (w.last_status & 0x400)


def test_limit_impulse_solves_the_lcp_with_the_mass_matrix():
    """One active limit row: x = -b / A with A = (M^-1)_dd, b = -qdot_pre, and v' = v_pre + M^-1 e_d x."""
    md = limited_arm()
    w = OracleWorld(md); n = w.n
    wf = OracleWorld(limited_arm(enforce=False))
    rng = np.random.default_rng(3)
    q = np.clip(rng.normal(0, 0.3, n), -0.15, 0.3); v = rng.normal(0, 0.5, n); a = rng.normal(0, 0.2, w.k)
    q[2] = 0.4; v[2] = 0.9                                                        # DOF 2 beyond its upper limit 0.4, moving outwards
    s = np.concatenate([q, v])
    nxt, vpre = w.step(s, a), wf.step(s, a)[n:]
    Minv = np.linalg.inv(w.mass_matrix(q))
    x = -vpre[2] / Minv[2, 2]
    assert x < 0 and np.allclose(nxt[n:], vpre + Minv[:, 2] * x, rtol=0, atol=1e-12)


def test_the_backward_pass_ignores_the_limit_row_like_the_references():
    """BackpropSnapshot builds A_c from DifferentiableContactConstraint::getConstraintForces, which is zero for a constraint that is not a
    contact: with a limit row clamping the Jacobians are those of the unconstrained step (the forward pass applied the impulse, the
    analytical backward pass does not know it - the reference's behaviour, restated as it is)."""
    md = limited_arm()
    w = OracleWorld(md); n = w.n
    wf = OracleWorld(limited_arm(enforce=False))
    rng = np.random.default_rng(4)
    q = np.clip(rng.normal(0, 0.3, n), -0.15, 0.3); v = rng.normal(0, 0.5, n); a = rng.normal(0, 0.2, w.k)
    q[1] = -0.6; v[1] = -0.8                                                      # below the lower limit -0.5
    s = np.concatenate([q, v]); g = rng.normal(0, 1, 2 * n)
    w.step(s, a); wf.step(s, a)
    assert w.last_status & 0x400
    gs, ga = w.backprop(g); gsf, gaf = wf.backprop(g)
    assert np.allclose(gs, gsf, rtol=0, atol=1e-10) and np.allclose(ga, gaf, rtol=0, atol=1e-10)


def test_limit_rows_share_the_lcp_with_contacts():
    """A limited arm resting on the ground with a joint at its limit: contact rows first, then the limit row, one LCP; the contact rows'
    part of the backward pass is still there (VJP = J^T g of the oracle's own step by central differences on the directions that do not
    involve the limit impulse is NOT expected - the reference's Jacobian ignores the limit row -, so compare with the oracle run
    without enforcement only where the limit row's impulse is zero)."""
    md = limited_arm(ground=True)
    w = OracleWorld(md); n = w.n
    wf = OracleWorld(limited_arm(ground=True, enforce=False))
    rng = np.random.default_rng(5)
    hit = 0
    for trial in range(40):
        q = np.clip(rng.normal(0, 0.2, n), -0.15, 0.3); v = rng.normal(0, 0.3, n); a = rng.normal(0, 0.2, w.k)
        q[0] = rng.uniform(-0.025, 0.005)                                        # the base's box on the ground in most trials
        q[2] = 0.4 + (rng.uniform(0, 0.05) if trial % 2 else 0.0)
        v[2] = rng.choice([-0.5, 0.8])
        s = np.concatenate([q, v])
        nxt = w.step(s, a)
        st = w.last_status
        assert st & 0x400
        if st & 0x1:
            hit += 1
        free = wf.step(s, a)
        if free[n + 2] < -1e-3:                    # still moving inwards after the contact impulses: the row's impulse is zero, same step as without it
            assert np.allclose(nxt, free, rtol=0, atol=1e-9)
        else:                                      # the contacts push the joint against its limit: the row stops it there
            assert nxt[n + 2] < 1e-5
    assert hit > 20


def test_the_lcp_cache_in_the_devices_slot_format_is_the_same_warm_start():
    """OracleWorld.set_lcp_cache_slots: the cache with three entries per constraint (a joint-limit row and a frictionless contact on the
    first of them) - the format the device keeps it in - is the same warm start as the reference's 3 / 1 / 1 rows: a second step from
    either gives bit-identical results, and the two caches hold the same numbers."""
    md = limited_arm(ground=True)
    for bx in md.boxes[1:2]:
        bx.mu = 0.0                                                             # one frictionless collider: a one-row contact
    w = OracleWorld(md); ws = OracleWorld(md); ws.set_lcp_cache_slots(True); n = w.n
    rng = np.random.default_rng(8)
    warm = 0
    for trial in range(30):
        q = np.clip(rng.normal(0, 0.2, n), -0.15, 0.3); v = rng.normal(0, 0.3, n); a = rng.normal(0, 0.2, w.k)
        q[0] = rng.uniform(-0.025, 0.005); q[2] = 0.4; v[2] = 0.8
        s = np.concatenate([q, v])
        w.reset_lcp_cache(); ws.reset_lcp_cache()
        s1 = w.step(s, a); s1s = ws.step(s, a)
        assert np.array_equal(s1, s1s)
        L = w.last_lcp(); c, cs = w.get_lcp_cache(), ws.get_lcp_cache()
        assert len(c) == len(L["x"])
        if len(cs) == 0:
            continue
        assert len(cs) % 3 == 0 and len(cs) >= len(c)
        assert np.array_equal(np.sort(np.abs(cs[cs != 0])), np.sort(np.abs(c[c != 0])))
        s2 = w.step(s1, a); st2 = w.last_status
        s2s = ws.step(s1s, a)
        assert np.array_equal(s2, s2s) and ws.last_status == st2
        warm += int(bool(st2 & 0x2))
        g = rng.normal(0, 1, 2 * n)
        assert all(np.array_equal(x, y) for x, y in zip(w.backprop(g), ws.backprop(g)))
    assert warm > 5
