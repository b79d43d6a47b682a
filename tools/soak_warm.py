"""Randomised soak of the WARM-STARTED step (GPU box): the soak's random models take one cold step, then a second step from the state
the device reached, the LCP started from the first step's solution (the reference's solver carries mX between steps,
BoxedLcpConstraintSolver.cpp:176-187).  The second step's next state and gradients against the oracle, which is given the same state and
the same first-step solution (the device's); worlds above 1e-5 must be proven reference-unstable (tools/soak_parity.py,
prove_reference_unstable: every probe with the same warm start).   usage: python tools/soak_warm.py [first seed] [count] [B] [mode: balls|multi|big] [stress mode of tools/soak_stress.py]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import nimblephysics_amd as na  # noqa: E402
import soak_parity  # noqa: E402
from nimblephysics_amd.timestep import timestep  # noqa: E402
from oracle import OracleWorld  # noqa: E402


def run(first=0, count=20, B=256, mode="balls", verbose=True, stress=None):
    """stress: a mode of tools/soak_stress.py applied to every model (joint-limit rows and frictionless contacts in the warm start: the
    device's cache and the oracle's are exchanged in the device's format, three entries per constraint - OracleWorld.set_lcp_cache_slots)."""
    tot = {"worlds": 0, "contact2": 0, "stage0": 0, "gt1e-7": 0, "gt1e-5": 0, "unstable": 0, "MISMATCH": 0}
    for seed in range(first, first + count):
        case = soak_parity.make_case(seed, B, big=mode == "big", multi=mode == "multi", balls=mode == "balls")
        if case is None:
            continue
        md, s, a, g = case
        if stress is not None:
            import soak_stress
            md, s, a, g = soak_stress.mutator(stress)(seed, md, s, a, g)
        try:
            world = na.World(md, device="cuda:0")
        except na.NimbleAmdError:
            continue
        ow = OracleWorld(md); ow.set_lcp_cache_slots(True)
        at = torch.tensor(a, device="cuda:0")
        with torch.no_grad():
            s1 = timestep(world, torch.tensor(s, device="cuda:0"), at)               # cold step; the world keeps its solution
        st1 = world.last_status.cpu().numpy().astype(np.uint32)
        r1 = ow.step_batch(s, a, None, threads=8, want_lcp=True)
        s1n = s1.cpu().numpy()
        ok1 = (np.abs(s1n - r1["next"]).max(1) <= 1e-7 * max(np.abs(r1["next"]).max(), 1e-30)) & (((st1 | r1["status"]) & 0x80) == 0)
        world_cache = world.lcp_cache.clone()
        st = s1.clone().requires_grad_(True); at2 = at.clone().requires_grad_(True)
        out = timestep(world, st, at2)                                                # warm step
        st2 = world.last_status.cpu().numpy().astype(np.uint32)
        out.backward(torch.tensor(g, device="cuda:0"))
        # the oracle starts from the DEVICE's first-step solution: on a rank-deficient A the two first-step solutions may differ in the
        # null space of A (same velocities, both valid), and the second step depends on which one it is given
        cache = world_cache.cpu().numpy()                                              # [25][B]: 24 impulses + the row count
        dev_lcp = np.ascontiguousarray(cache[:r1["lcp"].shape[1]].T); dev_len = cache[-1].astype(np.int32)
        same_rows = dev_len == r1["lcp_len"]
        r2 = ow.step_batch(s1n, a, g, threads=8, lcp_in=dev_lcp, lcp_len_in=dev_len)
        r1 = dict(r1, lcp=dev_lcp, lcp_len=dev_len)
        dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at2.grad.cpu().numpy()}
        scales = {k: max(np.abs(r2[k]).max(), 1e-30) for k in dev}
        err = np.maximum.reduce([np.abs(dev[k] - r2[k]).max(1) / scales[k] for k in dev])
        use = ok1 & same_rows & (((st2 | r2["status"]) & 0x80) == 0)
        for k in dev:                                          # (a NaN would compare as "not above the tolerance")
            assert np.isfinite(dev[k][use]).all() and np.isfinite(r2[k][use]).all(), ("non-finite result", seed, k)
        err[~use] = 0.0
        bad = np.where(err > 1e-5)[0]
        prng = np.random.default_rng(1)
        unstable = mismatch = 0
        dev_cache2 = world.lcp_cache.cpu().numpy()                                     # the second step's solution
        for wd in bad:
            how, spread, nearest = soak_parity.prove_reference_unstable(
                ow, seed, 1e-5, s1n[wd], a[wd], g[wd], {k: dev[k][wd] for k in dev}, {k: r2[k][wd] for k in dev}, scales, int(st2[wd]),
                dev_cache2[:, wd], prng, lcp=(r1["lcp"][wd], int(r1["lcp_len"][wd])))
            if how is not None:
                unstable += 1
                if how != "state":
                    tot[how] = tot.get(how, 0) + 1
            elif (soak_parity.near_log_map_singularity(md, r2["next"][wd]) and err[wd] < 3e-3
                  and max(np.abs(dev[k][wd] - r2[k][wd]).max() / scales[k] for k in ("next", "grad_action")) <= 1e-5):
                tot["reference_fd_near_pi"] = tot.get("reference_fd_near_pi", 0) + 1   # (the reference's finite-differenced SO(3) integration)
            elif soak_parity.exact_derivatives_agree(ow, 1e-5, s1n[wd], a[wd], g[wd], {k: dev[k][wd] for k in dev}, scales,
                                                     lcp=(r1["lcp"][wd], int(r1["lcp_len"][wd]))):
                tot["reference_fd_exact_agrees"] = tot.get("reference_fd_exact_agrees", 0) + 1   # (... proven by the exact-derivative instrument)
            else:
                mismatch += 1
                print(f"  MISMATCH seed {seed} world {wd}: err {err[wd]:.2e} spread {spread:.2e} nearest {nearest:.2e} status dev {st2[wd]:#x} ref {r2['status'][wd]:#x}")
        c = ((st2 & 1) != 0) & use
        tot["worlds"] += int(use.sum()); tot["contact2"] += int(c.sum()); tot["stage0"] += int((c & ((st2 & 2) != 0)).sum())
        tot["gt1e-7"] += int((err > 1e-7).sum()); tot["gt1e-5"] += len(bad); tot["unstable"] += unstable; tot["MISMATCH"] += mismatch
        if verbose and len(bad):
            print(f"seed {seed}: used {use.sum()} in contact {c.sum()} >1e-7 {(err > 1e-7).sum()} >1e-5 {len(bad)} (unstable {unstable}, mismatch {mismatch})", flush=True)
    return tot


if __name__ == "__main__":
    print(run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 20, int(sys.argv[3]) if len(sys.argv) > 3 else 256,
              sys.argv[4] if len(sys.argv) > 4 else "balls", stress=sys.argv[5] if len(sys.argv) > 5 else None))
