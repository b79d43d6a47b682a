"""Shared parity criterion of the GPU tests: every world within `tol` of the oracle, or PROOF that the reference algorithm has no
stable answer on that world (tests/test_gpu_contact.py states the reasoning).  Test infrastructure.

The error norm is PER WORLD AND PER BLOCK: the next state splits into q' and v', the state gradient into the position
and the velocity cotangent, the action gradient is one block; an entry's error is measured against the magnitude of ITS block in ITS
world of the reference (max |.| over the block), with a floor of FLOOR x the batch-wide magnitude of that block so that a block that
happens to be ~0 in one world is not held to an absolute accuracy nothing has.  (Round 3 divided by the batch-wide maximum of the
whole output: velocities of 1e-2 next to q0 = pi/2 were held to 1.6e-5 of themselves at tol 1e-7.)"""
import numpy as np

KEYS = ("next", "grad_state", "grad_action")
EPS = 2.220446049250313e-16
FLOOR = 1e-2


def _blocks(key, dim):
    """index ranges of the blocks of one output: next = [q'; v'], grad_state = [dL/dq; dL/dv], grad_action = one block"""
    if key == "grad_action" or dim % 2:
        return [(0, dim)]
    return [(0, dim // 2), (dim // 2, dim)]


def entry_scales(ref):
    """-> {key: [B, dim] array}: for every entry the magnitude it is measured against (its world's block of the reference, floored)."""
    out = {}
    for k in KEYS:
        r = np.abs(np.asarray(ref[k]))
        sc = np.empty_like(r)
        for lo, hi in _blocks(k, r.shape[1]):
            if hi <= lo:
                continue
            per_world = r[:, lo:hi].max(1)
            floor = max(FLOOR * float(per_world.max()), 1e-30)
            sc[:, lo:hi] = np.maximum(per_world, floor)[:, None]
        out[k] = sc
    return out


def block_errors(x, ref, nblocks=1):
    """The same norm for one array [B, dim] split into `nblocks` equal blocks: per world, max over its entries of |x - ref| / (the
    magnitude of the entry's block in that world of `ref`, floored at FLOOR x the batch-wide magnitude of the block)."""
    x, ref = np.asarray(x), np.asarray(ref)
    B, dim = ref.shape
    out = np.zeros(B)
    w = dim // nblocks
    for b in range(nblocks):
        lo, hi = b * w, (dim if b == nblocks - 1 else (b + 1) * w)
        pw = np.abs(ref[:, lo:hi]).max(1)
        sc = np.maximum(pw, max(FLOOR * float(pw.max()), 1e-30))
        out = np.maximum(out, np.abs(x[:, lo:hi] - ref[:, lo:hi]).max(1) / sc)
    return out


def world_errors(dev, ref):
    """Per-world error of every output: max over its entries of |dev - ref| / (the entry's block magnitude in that world, floored).
    -> ({key: [B]}, {key: [B, dim] entry scales})."""
    scales = entry_scales(ref)
    return {k: (np.abs(dev[k] - ref[k]) / scales[k]).max(1) if dev[k].shape[1] else np.zeros(len(dev[k])) for k in KEYS}, scales


def log_map_gap(md, next_state):
    """Per world: how far the NEXT rotation angle of its exponential-map joints (free, ball) is from pi (inf without such joints).  The
    reference finite-differences the position integration of those joints (central differences, eps 1e-6: FreeJoint.cpp:950-1007,
    BallJoint.cpp:351-408; the oracle restates that) and its posPos / velPos blocks lose digits like 1 / gap^2 towards the log-map
    singularity; the device differentiates exactly (tests/test_gpu_ball_joint.py pins it against 80-bit arithmetic)."""
    next_state = np.asarray(next_state)
    gap = np.full(len(next_state), np.inf)
    off = 0
    for b in md.bodies:
        nd = {"free": 6, "weld": 0, "ball": 3}.get(b.joint_type, 1)
        if b.joint_type in ("free", "ball"):
            gap = np.minimum(gap, np.abs(np.linalg.norm(next_state[:, off:off + 3], axis=1) - np.pi))
        off += nd
    return gap


FD_COEF, FD_GAP, FD_CAP = 5e-9, 0.15, 3e-3


def gradient_tolerance(md, next_state, tol):
    """Per-world tolerance of the GRADIENT blocks: `tol`, except within FD_GAP of the log-map singularity of a free / ball joint, where the
    reference's own finite-difference Jacobian is only good to ~FD_COEF / gap^2 (measured on cfg4: 2e-7 at 0.05 rad, 4e-5 at 0.004)."""
    gap = log_map_gap(md, next_state)
    with np.errstate(divide="ignore"):
        fd = np.minimum(FD_COEF / np.maximum(gap, 1e-12) ** 2, FD_CAP)
    return np.where(gap < FD_GAP, np.maximum(tol, fd), tol)


def assert_match_or_reference_unstable(tag, ow, s, a, g, dev, ref, tol, lcp=None, lcp_len=None, n_perturb=64, closeness=0.1, ulps=1,
                                       max_unstable=None, max_by_closeness=None, fd_model=None, only=None):
    """dev / ref: dicts of next, grad_state, grad_action [B, .].  Worlds above `tol` must be ones where the oracle's OWN result moves by
    more than `tol` under +-`ulps`-ulp perturbations of its inputs (state, and the LCP warm start when one is given), and the device
    result must be one of the oracle's outcomes: within `tol` of a perturbed run ("tol" branch), or - where those outcomes form a
    continuum - at least 1 / `closeness` times closer to one of them than they scatter ("closeness" branch).  The closeness branch is
    BOUNDED: at most `max_by_closeness` worlds may need it (default: 2 worlds or 0.5 % of the batch).  fd_model: the model description -
    the gradients of worlds next to the log-map singularity of a free / ball joint are then held to the accuracy of the reference's own
    finite differences there (gradient_tolerance), everything else to `tol`.  only: a mask of the worlds to judge.  Prints how many
    worlds took which branch and returns (unstable worlds, worlds that needed the closeness branch)."""
    for k in KEYS:
        assert np.isfinite(dev[k]).all() and np.isfinite(ref[k]).all(), (tag, k, "non-finite values", int((~np.isfinite(dev[k])).sum()), int((~np.isfinite(ref[k])).sum()))
    errs, scales = world_errors(dev, ref)
    B = len(errs["next"])
    tolg = gradient_tolerance(fd_model, ref["next"], tol) if fd_model is not None else np.full(B, tol)
    tols = {"next": np.full(B, tol), "grad_state": tolg, "grad_action": tolg}
    excess = np.maximum.reduce([errs[k] / tols[k] for k in KEYS])           # > 1: above the tolerance of that output in that world
    worst = np.maximum.reduce([errs[k] for k in KEYS])
    if only is not None:
        excess = np.where(only, excess, 0.0)
    bad = np.where(excess > 1.0)[0]
    rng = np.random.default_rng(12345)
    by_tol = by_closeness = 0
    for wd in bad:
        s0 = s[wd]
        sp = s0[None, :] * (1.0 + rng.integers(-ulps, ulps + 1, (n_perturb, s0.size)) * EPS)
        kw = {}
        if lcp is not None:
            kw = {"lcp_in": np.repeat(lcp[wd][None], n_perturb, 0) * (1.0 + rng.integers(-ulps, ulps + 1, (n_perturb, lcp.shape[1])) * EPS),
                  "lcp_len_in": np.repeat(lcp_len[wd], n_perturb)}
        r = ow.step_batch(sp, np.repeat(a[wd][None], n_perturb, 0), np.repeat(g[wd][None], n_perturb, 0), threads=8, **kw)
        nz = [k for k in KEYS if dev[k].shape[1]]
        tw = float(tolg[wd])
        dist = np.maximum.reduce([(np.abs(r[k] - dev[k][wd][None]) / scales[k][wd][None]).max(1) for k in nz])
        spread = max(float((np.abs(r[k] - ref[k][wd][None]) / scales[k][wd][None]).max()) for k in nz)
        assert spread > tw, (tag, int(wd), "the reference is stable here but the device differs", float(worst[wd]), float(dist.min()), tw)
        assert dist.min() <= max(tw, closeness * spread), (tag, int(wd), "device result is none of the reference's own outcomes",
                                                           float(dist.min()), float(spread))
        if dist.min() <= tw:
            by_tol += 1
        else:
            by_closeness += 1
    near_pi = int((tolg > tol).sum())
    print(f"[{tag}] worlds above 1e-7 / above {tol:g} (per world and block): {(worst > 1e-7).sum()} / {(worst > tol).sum()} of {B} (max {worst.max():.2e})"
          + (f"; {near_pi} worlds next to a log-map singularity held to the reference's finite-difference accuracy" if near_pi else "")
          + f"; reference-unstable (oracle flips under {ulps}-ulp perturbations): {len(bad)}, device within tol of one of its outcomes: {by_tol}, "
          f"accepted by the closeness branch: {by_closeness}")
    if max_unstable is not None:
        assert len(bad) <= max_unstable, (tag, len(bad), max_unstable)
    cap = max(2, int(0.005 * B)) if max_by_closeness is None else max_by_closeness
    assert by_closeness <= cap, (tag, "too many worlds needed the closeness branch", by_closeness, cap)
    return len(bad), by_closeness
