"""The oracle's test instrument `set_exact_position_jacobians` (oracle/dynamics.hpp::posJacobiansExact): the position-integration
Jacobians of free and ball joints by forward-mode differentiation of expMapRot / logMap instead of the reference's central differences
(FreeJoint.cpp:950-1007, BallJoint.cpp:351-408).  Pinned here, on the CPU, against a five-point stencil of the same function in 80-bit
arithmetic: next to the log-map singularity the instrument is exact where the reference's quotient is off by 1e-6 .. 1e-3, away from it
the two agree - which is what tests/parity.py relies on when it judges the gradients of worlds next to the singularity against the
oracle WITH the instrument (and what tests/test_gpu_contact.py::test_cfg4_box_stack_8192_worlds then shows on the device)."""
import numpy as np

from oracle import OracleWorld


def _stencil(q, w, dt, g):
    from test_gpu_ball_joint import _so3_vjp_extended_precision
    # h = 1e-4: the true function is smooth up to pi (the 1 / gap^2 is the conditioning of logMap's FORMULA), so the truncation error of the
    # five-point stencil is nothing, while its rounding error - 1e-19 / gap^2 of the 80-bit evaluation divided by h - is what limits it:
    # against the instrument h = 1e-6 / 1e-5 / 1e-4 gives 7e-8 / 1e-8 / 9e-10 at gap = 2e-3
    return _so3_vjp_extended_precision(q, w, dt, g, h=1e-4)


def _cube_states(B, seed, gap_lo, gap_hi):
    """cfg4's world (two free-joint cubes on the ground) with the LOWER cube's rotation vector `gap` short of pi in norm"""
    from util import box_stack_inputs
    md, s, a = box_stack_inputs(B, seed)
    rng = np.random.default_rng(seed + 100)
    n = md.num_dofs
    ax = rng.normal(size=(B, 3)); ax[:, [0, 2]] *= 0.05; ax /= np.linalg.norm(ax, axis=1, keepdims=True)   # close to a yaw (the cube stays flat)
    s[:, 0:3] = ax * (np.pi - rng.uniform(gap_lo, gap_hi, (B, 1)))
    s[:, n:n + 3] = rng.normal(0, 0.5, (B, 3))
    return md, s, a


def _rot_block_errors(md, s, a, exact):
    """max relative error of the rotation blocks of posPos^T g / velPos^T g of the first free joint against the 80-bit stencil, per world.
    A cotangent on the next ROTATION of that joint only: then dL/dq[0:3] = posPos_rr^T g and dL/dv[0:3] = velPos_rr^T g exactly (the next
    positions do not depend on anything else), contacts or not."""
    B, n = len(s), md.num_dofs
    rng = np.random.default_rng(7)
    g = np.zeros((B, 2 * n)); g[:, 0:3] = rng.normal(0, 1, (B, 3))
    ow = OracleWorld(md)
    ow.set_exact_position_jacobians(exact)
    gs = ow.step_batch(s, a, g, threads=4)["grad_state"]
    err = np.zeros(B)
    for w in range(B):
        x = _stencil(s[w, 0:3], s[w, n:n + 3], md.dt, g[w, 0:3])
        got = np.concatenate([gs[w, 0:3], gs[w, n:n + 3]])
        err[w] = np.abs(got - x).max() / np.abs(x).max()
    return err


def test_exact_position_jacobians_next_to_the_log_map_singularity():
    md, s, a = _cube_states(48, 3, 2e-3, 2e-2)
    e_fd = _rot_block_errors(md, s, a, False)
    e_ex = _rot_block_errors(md, s, a, True)
    print(f"rotation 2e-3 .. 2e-2 rad short of pi, vs the 80-bit stencil: central differences (the reference) max {e_fd.max():.1e} "
          f"median {np.median(e_fd):.1e}; exact instrument max {e_ex.max():.1e}")
    assert e_ex.max() < 1e-8, e_ex.max()
    assert e_fd.max() > 1e-6 and np.median(e_fd) > 30 * e_ex.max()


def test_exact_and_finite_difference_jacobians_agree_away_from_the_singularity():
    """... and everywhere else the instrument changes nothing that a test at 1e-7 could see: whole-state gradients of cfg4 worlds (contacts,
    cascade and all) with the rotation at least 0.5 rad from pi agree to 5e-8 between the two modes; the forward pass is untouched."""
    from util import box_stack_inputs
    md, s, a = box_stack_inputs(256, 11)
    n = md.num_dofs
    far = (np.pi - np.abs(np.linalg.norm(s[:, 0:3], axis=1)) > 0.5) & (np.pi - np.abs(np.linalg.norm(s[:, 6:9], axis=1)) > 0.5)
    assert far.sum() > 100
    g = np.random.default_rng(5).normal(0, 1, s.shape)
    ow = OracleWorld(md)
    r0 = ow.step_batch(s, a, g, threads=4)
    ow.set_exact_position_jacobians(True)
    r1 = ow.step_batch(s, a, g, threads=4)
    ow.set_exact_position_jacobians(False)
    r2 = ow.step_batch(s, a, g, threads=4)
    assert np.array_equal(r0["next"], r1["next"]) and np.array_equal(r0["grad_action"], r1["grad_action"])
    assert all(np.array_equal(r0[k], r2[k]) for k in ("next", "grad_state", "grad_action"))      # switched off: the reference again
    sc = np.abs(r0["grad_state"]).max(1, keepdims=True)
    d = (np.abs(r1["grad_state"] - r0["grad_state"]) / sc).max(1)
    print(f"exact vs central differences, {int(far.sum())} worlds >= 0.5 rad from pi: max {d[far].max():.1e}; the other {int((~far).sum())}: max {d[~far].max():.1e}")
    assert d[far].max() < 5e-8      # (the quotient itself: ~5e-9 / gap^2 = 2e-8 at 0.5 rad)


def test_exact_position_jacobians_of_a_ball_joint():
    from test_gpu_ball_joint import ball_model
    from test_ball_joint import _ball_offsets
    md = ball_model(10, False, properties=False)
    n = md.num_dofs; B = 16
    rng = np.random.default_rng(90)
    q = rng.normal(0, 0.4, (B, n)); v = rng.normal(0, 0.5, (B, n))
    offs = _ball_offsets(md)
    for o in offs:
        ax = rng.normal(size=(B, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
        q[:, o:o + 3] = ax * (np.pi - rng.uniform(2e-3, 2e-2, (B, 1)))
    s = np.concatenate([q, v], 1); a = np.zeros((B, len(md.action_map)))
    g = np.concatenate([rng.normal(0, 1, (B, n)), np.zeros((B, n))], 1)
    ow = OracleWorld(md)
    out = {}
    for exact in (False, True):
        ow.set_exact_position_jacobians(exact)
        gs = ow.step_batch(s, a, g, threads=4)["grad_state"]
        e = 0.0
        for w in range(B):
            for o in offs:
                x = _stencil(q[w, o:o + 3], v[w, o:o + 3], md.dt, g[w, o:o + 3])
                idx = list(range(o, o + 3)) + list(range(n + o, n + o + 3))
                e = max(e, np.abs(gs[w, idx] - x).max() / np.abs(x).max())
        out[exact] = e
    print(f"ball joints next to pi vs the 80-bit stencil: central differences {out[False]:.1e}, exact instrument {out[True]:.1e}")
    assert out[True] < 1e-8 and out[False] > 30 * out[True]
