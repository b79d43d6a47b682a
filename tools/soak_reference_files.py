"""Every SKEL world and URDF robot of the reference's data directory that the loaders accept, stepped forward + backward on the device and in the
oracle on random states (GPU box, needs /root/reference - a development tool, not part of the suite): loader -> description -> device
against loader -> description -> oracle, so what it pins is the DEVICE on the reference's own model files (joint types, frames, inertias,
collider placement), not the loader.   usage: python tools/soak_reference_files.py [dir] [B]"""
import glob
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


def run(directory, B=64, tol=1e-6):
    tot = {"files": 0, "loaded": 0, "refused_by_loader": 0, "refused_by_device": 0, "worlds": 0, "contact": 0, "gt_tol": 0, "unstable": 0, "MISMATCH": 0}
    for path in sorted(glob.glob(os.path.join(directory, "**", "*.skel"), recursive=True) + glob.glob(os.path.join(directory, "**", "*.urdf"), recursive=True)):
        tot["files"] += 1
        try:
            # (URDF: colliders outside the analytic narrow phases - meshes - are dropped: what is compared is the dynamics of the file's tree)
            md = na.load_skel(path) if path.endswith(".skel") else na.load_urdf(path, drop_unsupported_colliders=True)
        except Exception:
            tot["refused_by_loader"] += 1
            continue
        if md.num_dofs == 0:
            continue
        if md.boxes and not md.max_contacts:
            md.max_contacts = md.suggest_max_contacts()
        try:
            world = na.World(md, device="cuda:0")
        except na.NimbleAmdError as e:
            tot["refused_by_device"] += 1
            print(f"  {os.path.relpath(path, directory)}: device refuses: {str(e)[:110]}")
            continue
        world.ref_layout = None      # (states in the DEVICE's layout here: the frozen coordinates of immobile skeletons are not part of the comparison)
        tot["loaded"] += 1
        rng = np.random.default_rng(abs(hash(os.path.basename(path))) % (2 ** 31))
        n = md.num_dofs; fl = md.flat()
        q = rng.normal(0, 0.2, (B, n)); v = rng.normal(0, 0.5, (B, n)); a = rng.normal(0, 0.5, (B, len(md.action_map))); g = rng.normal(0, 1, (B, 2 * n))
        lo, hi = np.asarray(fl["pos_lo"], dtype=np.float64), np.asarray(fl["pos_hi"], dtype=np.float64)
        q = np.clip(q, np.where(np.isfinite(lo), lo + 1e-3, -np.inf), np.where(np.isfinite(hi), hi - 1e-3, np.inf))
        s = np.concatenate([q, v], 1)
        ow = OracleWorld(md)
        st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
        out = timestep(world, st, at)
        status = world.last_status.cpu().numpy().astype(np.uint32)
        dev_cache = world.lcp_cache.cpu().numpy() if world.lcp_cache is not None else np.zeros((3 * max(md.max_contacts, 8) + 1, B))
        out.backward(torch.tensor(g, device="cuda:0"))
        ref = ow.step_batch(s, a, g, threads=8)
        dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
        for k in dev:
            assert np.isfinite(dev[k]).all() and np.isfinite(ref[k]).all(), (path, k)
        scales = {k: max(np.abs(ref[k]).max(), 1e-30) for k in dev}
        err = np.maximum.reduce([np.abs(dev[k] - ref[k]).max(1) / scales[k] for k in dev])
        err[((status | ref["status"]) & 0x80) != 0] = 0.0
        prng = np.random.default_rng(1)
        for wd in np.where(err > tol)[0]:
            how, spread, nearest = soak_parity.prove_reference_unstable(ow, 0, tol, s[wd], a[wd], g[wd], {k: dev[k][wd] for k in dev}, {k: ref[k][wd] for k in dev},
                                                                        scales, int(status[wd]), dev_cache[:, wd], prng)
            if how is not None:
                tot["unstable"] += 1
            else:
                tot["MISMATCH"] += 1
                print(f"  MISMATCH {os.path.relpath(path, directory)} world {wd}: err {err[wd]:.2e} status dev {status[wd]:#x} ref {ref['status'][wd]:#x}")
        tot["worlds"] += B; tot["contact"] += int((status & 1).sum()); tot["gt_tol"] += int((err > tol).sum())
        print(f"  {os.path.relpath(path, directory)}: n {n} bodies {len(md.bodies)} colliders {len(md.boxes)} contact {int((status & 1).sum())} max err {err.max():.1e}", flush=True)
    return tot


if __name__ == "__main__":
    print(run(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data/skel", int(sys.argv[2]) if len(sys.argv) > 2 else 64))
