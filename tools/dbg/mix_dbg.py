"""One case of the mixed-feature soak (tools/soak_stress.py mix): which mutations, statuses, contacts of the differing worlds.
usage (GPU box): python tools/dbg/mix_dbg.py <seed> <balls|big|multi> [world]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
import soak_parity, soak_stress
seed = int(sys.argv[1]); variant = sys.argv[2]; B = 256
mode = sys.argv[4] if len(sys.argv) > 4 else "mix"
pick = np.random.default_rng(seed + 77).random(len(soak_stress.MIX_ORDER)) < 0.5
print("parts", [m for m, p in zip(soak_stress.MIX_ORDER, pick) if p] if mode == "mix" else mode)
md, s, a, g = soak_parity.make_case(seed, B, variant == "big", variant == "multi", variant == "balls", False)
md, s, a, g = soak_stress.mutator(mode)(seed, md, s, a, g)
n = md.num_dofs
print("nb", len(md.bodies), "n", n, "colliders", [(bx.shape, bx.body) for bx in md.boxes], "limits", [i for i, b in enumerate(md.bodies) if b.limit_enforced],
      "selfcol", any(b.self_collision for b in md.bodies), "skeletons", md.body_skeletons() if hasattr(md, "body_skeletons") else None)
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
out = timestep(world, st, at)
status = world.last_status.cpu().numpy().astype(np.uint32)
out.backward(torch.tensor(g, device="cuda:0"))
ref = ow.step_batch(s, a, g, threads=8)
dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
scales = {k: max(np.abs(ref[k]).max(), 1e-30) for k in dev}
errs = {k: np.abs(dev[k] - ref[k]).max(1) / scales[k] for k in dev}
err = np.maximum.reduce(list(errs.values()))
d = np.where((status & 0x481) != (ref["status"] & 0x481))[0]
print("flag differences:", d, [hex(x) for x in status[d]], [hex(x) for x in ref["status"][d]])
bad = np.where(err > 1e-6)[0]
print("err>1e-6:", [(int(w), {k: float(f"{errs[k][w]:.2e}") for k in errs}, hex(status[w]), hex(ref["status"][w])) for w in bad])
np.set_printoptions(linewidth=220, precision=6)
worlds = [int(sys.argv[3])] if len(sys.argv) > 3 and int(sys.argv[3]) >= 0 else list(d[:2]) + list(bad[:2])
for wd in worlds:
    ow.step(s[wd], a[wd])
    cts = ow.last_contacts()
    print(f"world {wd}: oracle {len(cts)} contacts, status {ref['status'][wd]:#x}; device status {status[wd]:#x}")
    for c in cts:
        print("   p", c[0:3], "n", c[3:6], "depth", c[6], "type", c[7], "rest", c[8:12])
    lc = ow.last_lcp() if hasattr(ow, "last_lcp") else None
    if lc is not None:
        print("   oracle lcp x", lc.get("x"), "\n   lo", lc.get("lo"), "\n   hi", lc.get("hi"), "\n   b", lc.get("b"))
    dbg = world.debug_contacts(wd) if hasattr(world, "debug_contacts") else None
    if dbg is not None:
        print("   device", dbg)
    import collections
    for ulps, absolute in ((1, False), (4, False), (1, True)):
        ow.set_lcp_noise(ulps, seed, absolute)
        nd = 256
        r = ow.step_batch(np.repeat(s[wd][None], nd, 0), np.repeat(a[wd][None], nd, 0), np.repeat(g[wd][None], nd, 0), threads=8)
        ow.set_lcp_noise(0)
        dk = {k: np.abs(r[k] - dev[k][wd][None]).max(1) / scales[k] for k in dev}
        same = r["status"] == status[wd]
        print(f"   noise {ulps} {'abs' if absolute else 'rel'}: statuses {dict(collections.Counter(hex(x) for x in r['status']))}; draws with the device's status: {int(same.sum())}")
        if same.any():
            i = np.argmin(np.maximum.reduce([dk[k] for k in dk]) + (~same) * 1e9)
            print("      nearest of them:", {k: float(f"{dk[k][i]:.2e}") for k in dk})
        if same.any() and os.environ.get("REPLAY"):
            ow.set_lcp_noise(ulps, seed, absolute)
            for _ in range(int(i) + 1):
                ow.reset_lcp_cache()
                ow.step(s[wd], a[wd])
            Lr = ow.last_lcp()
            gq = ow.backprop(g[wd]) if hasattr(ow, "backprop") else None
            ow.set_lcp_noise(0)
            print("      replayed draw", int(i), "status", hex(ow.last_status), "\n      x", Lr["x"], "\n      classes", Lr["row_class"], "\n      mu x_n", [Lr["hi"][k] * Lr["x"][Lr["findex"][k]] if Lr["findex"][k] >= 0 else None for k in range(len(Lr["x"]))])
    dc = world.lcp_cache.cpu().numpy()
    ln = int(dc[-1, wd])
    r = ow.step_batch(s[wd][None], a[wd][None], g[wd][None], threads=1, lcp_in=np.ascontiguousarray(dc[:-1, wd][None]), lcp_len_in=np.array([ln], np.int32), want_lcp=True)
    print("   replay of the device's solution", dc[:ln, wd], "-> oracle status", hex(r["status"][0]), {k: float(f"{np.abs(r[k][0] - dev[k][wd]).max() / scales[k]:.2e}") for k in dev},
          "\n   oracle's solution after it", r["lcp"][0][:ln])
    ow.reset_lcp_cache(); ow.step(s[wd], a[wd]); Lr = ow.last_lcp(); nct = len(ow.last_contacts())
    rows_of, r_, c_ = [], 0, 0
    while r_ < len(Lr["b"]):
        k3 = c_ < nct and r_ + 2 < len(Lr["b"]) and Lr["findex"][r_ + 1] == r_ and Lr["findex"][r_ + 2] == r_
        rows_of += [3 * c_, 3 * c_ + 1, 3 * c_ + 2] if k3 else [3 * c_]
        r_ += 3 if k3 else 1; c_ += 1
    print("   device rows", ln, "oracle constraints", c_, "-> forced x", dc[rows_of, wd])
    ow.reset_lcp_cache(); ow.set_lcp_forced(dc[rows_of, wd])
    nx = ow.step(s[wd], a[wd]); stf = ow.last_status; gs, ga = ow.backprop(g[wd]); ow.set_lcp_forced(None)
    print("   forced replay: status", hex(stf), "next", np.abs(nx - dev["next"][wd]).max() / scales["next"], "grad_state", np.abs(gs - dev["grad_state"][wd]).max() / scales["grad_state"],
          "grad_action", np.abs(ga - dev["grad_action"][wd]).max() / scales["grad_action"])
