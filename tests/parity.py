"""Shared parity criterion of the GPU tests: every world within `tol` of the oracle, or PROOF that the reference algorithm has no
stable answer on that world (tests/test_gpu_contact.py states the reasoning).  Test infrastructure."""
import numpy as np

KEYS = ("next", "grad_state", "grad_action")
EPS = 2.220446049250313e-16


def world_errors(dev, ref):
    """Per-world max error of every output, relative to the batch-wide magnitude of that output (north_star: 1e-5 relative)."""
    scales = {k: max(float(np.abs(ref[k]).max()), 1e-30) for k in KEYS}
    return {k: np.abs(dev[k] - ref[k]).max(1) / scales[k] for k in KEYS}, scales


def assert_match_or_reference_unstable(tag, ow, s, a, g, dev, ref, tol, lcp=None, lcp_len=None, n_perturb=64, closeness=0.1, ulps=1,
                                       max_unstable=None):
    """dev / ref: dicts of next, grad_state, grad_action [B, .].  Worlds above `tol` must be ones where the oracle's OWN result moves by
    more than `tol` under +-`ulps`-ulp perturbations of its inputs (state, and the LCP warm start when one is given), and the device
    result must be one of the oracle's outcomes: within `tol` of a perturbed run ("tol" branch), or - where those outcomes form a
    continuum - at least 1 / `closeness` times closer to one of them than they scatter ("closeness" branch).  Prints how many worlds
    took which branch and returns (unstable worlds, worlds that needed the closeness branch)."""
    for k in KEYS:
        assert np.isfinite(dev[k]).all() and np.isfinite(ref[k]).all(), (tag, k, "non-finite values", int((~np.isfinite(dev[k])).sum()), int((~np.isfinite(ref[k])).sum()))
    errs, scales = world_errors(dev, ref)
    worst = np.maximum.reduce([errs[k] for k in KEYS])
    bad = np.where(worst > tol)[0]
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
        dist = np.maximum.reduce([np.abs(r[k] - dev[k][wd][None]).max(1) / scales[k] for k in KEYS])
        spread = max(np.abs(r[k] - ref[k][wd][None]).max() / scales[k] for k in KEYS)
        assert spread > tol, (tag, int(wd), "the reference is stable here but the device differs", float(worst[wd]), float(dist.min()))
        assert dist.min() <= max(tol, closeness * spread), (tag, int(wd), "device result is none of the reference's own outcomes",
                                                            float(dist.min()), float(spread))
        if dist.min() <= tol:
            by_tol += 1
        else:
            by_closeness += 1
    print(f"[{tag}] worlds above 1e-7 / above {tol:g}: {(worst > 1e-7).sum()} / {len(bad)} of {len(worst)} (max {worst.max():.2e}); "
          f"reference-unstable (oracle flips under {ulps}-ulp perturbations): {len(bad)}, device within tol of one of its outcomes: {by_tol}, "
          f"accepted by the closeness branch: {by_closeness}")
    if max_unstable is not None:
        assert len(bad) <= max_unstable, (tag, len(bad), max_unstable)
    return len(bad), by_closeness
