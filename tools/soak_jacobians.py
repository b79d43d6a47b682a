"""Randomised soak of the DENSE step Jacobians (GPU box): for the soak's random models, the device's d next_state / d state and
d next_state / d action (2n vector-Jacobian products each, nimblephysics_amd.neural.BackpropSnapshot.getStateJacobian / getActionJacobian)
against the oracle's World::getStateJacobian / getActionJacobian, a few worlds per model.  Every column of every block is compared, not one
random cotangent.   usage: python tools/soak_jacobians.py [first seed] [count] [worlds per model] [mode: balls|multi|big]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import nimblephysics_amd as na  # noqa: E402
import soak_parity  # noqa: E402
from oracle import OracleWorld  # noqa: E402


def run(first=0, count=20, W=4, mode="balls", verbose=True, stress=None):
    tot = {"models": 0, "worlds": 0, "contact": 0, "gt1e-7": 0, "gt1e-5": 0, "worst": 0.0}
    for seed in range(first, first + count):
        case = soak_parity.make_case(seed, 64, big=mode == "big", multi=mode == "multi", balls=mode == "balls")
        if case is None:
            continue
        md, s, a, g_ = case
        if stress is not None:
            import soak_stress
            md, s, a, g_ = soak_stress.mutator(stress)(seed, md, s, a, g_)
        s, a = s[:W], a[:W]
        try:
            world = na.World(md, device="cuda:0")
        except na.NimbleAmdError:
            continue
        world.setState(torch.tensor(s)); world.setAction(torch.tensor(a))
        snap = na.neural.forwardPass(world, idempotent=True)
        st = snap.getStatus().cpu().numpy().astype(np.uint32)
        Js, Ja = snap.getStateJacobian(world).cpu().numpy(), snap.getActionJacobian(world).cpu().numpy()
        ow = OracleWorld(md)
        for b in range(W):
            ow.reset_lcp_cache()        # every world of the device batch starts cold; the oracle world would carry world b-1's solution as a warm start
            ow.step(s[b], a[b])
            if (st[b] | ow.last_status) & 0x80:
                continue
            Rs, Ra = ow.getStateJacobian(), ow.getActionJacobian()
            assert np.isfinite(Js[b]).all() and np.isfinite(Ja[b]).all() and np.isfinite(Rs).all() and np.isfinite(Ra).all(), ("non-finite Jacobian", seed, b)
            e = max(np.abs(Js[b] - Rs).max() / max(np.abs(Rs).max(), 1e-30), np.abs(Ja[b] - Ra).max() / max(np.abs(Ra).max(), 1e-30))
            tot["worlds"] += 1; tot["contact"] += int(st[b] & 1); tot["gt1e-7"] += int(e > 1e-7); tot["gt1e-5"] += int(e > 1e-5)
            tot["worst"] = max(tot["worst"], float(e))
            if e > 1e-5 and verbose:
                print(f"  seed {seed} world {b}: Jacobian err {e:.2e} status dev {st[b]:#x} ref {ow.last_status:#x}")
        tot["models"] += 1
    return tot


if __name__ == "__main__":
    print(run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 20, int(sys.argv[3]) if len(sys.argv) > 3 else 4,
              sys.argv[4] if len(sys.argv) > 4 else "balls", stress=sys.argv[5] if len(sys.argv) > 5 else None))
