"""GPU: one world of the warm-start soak (tools/soak_warm.py): cold step, then the warm step on device and oracle.
usage: python tools/dbg/warm_dbg.py <seed> <world> <variant> [stress]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
import soak_parity, soak_stress
seed, wd, variant = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]; stress = sys.argv[4] if len(sys.argv) > 4 else None
B = 256
md, s, a, g = soak_parity.make_case(seed, B, variant == "big", variant == "multi", variant == "balls", False)
if stress:
    if stress == "mix":
        pick = np.random.default_rng(seed + 77).random(len(soak_stress.MIX_ORDER)) < 0.5
        print("parts", [m for m, p in zip(soak_stress.MIX_ORDER, pick) if p])
    md, s, a, g = soak_stress.mutator(stress)(seed, md, s, a, g)
np.set_printoptions(linewidth=220, precision=8)
print("nb", len(md.bodies), "n", md.num_dofs, "joints", [b.joint_type for b in md.bodies], "limits", [i for i, b in enumerate(md.bodies) if b.limit_enforced], "dt", md.dt)
world = na.World(md, device="cuda:0"); ow = OracleWorld(md); ow.set_lcp_cache_slots(True)
at = torch.tensor(a, device="cuda:0")
with torch.no_grad():
    s1 = timestep(world, torch.tensor(s, device="cuda:0"), at)
st1 = world.last_status.cpu().numpy().astype(np.uint32)
r1 = ow.step_batch(s, a, None, threads=8, want_lcp=True)
s1n = s1.cpu().numpy()
print("step 1: status dev", hex(st1[wd]), "ref", hex(r1["status"][wd]), "next err", np.abs(s1n[wd] - r1["next"][wd]).max())
cache = world.lcp_cache.cpu().numpy()
print("   device cache", cache[:int(cache[-1, wd]), wd], "rows", cache[-1, wd]); print("   oracle cache", r1["lcp"][wd][:r1["lcp_len"][wd]], "rows", r1["lcp_len"][wd])
st = s1.clone().requires_grad_(True); at2 = at.clone().requires_grad_(True)
out = timestep(world, st, at2)
st2 = world.last_status.cpu().numpy().astype(np.uint32)
out.backward(torch.tensor(g, device="cuda:0"))
dev_lcp = np.ascontiguousarray(cache[:r1["lcp"].shape[1]].T); dev_len = cache[-1].astype(np.int32)
r2 = ow.step_batch(s1n[wd][None], a[wd][None], g[wd][None], threads=1, lcp_in=dev_lcp[wd][None], lcp_len_in=dev_len[wd:wd + 1], want_lcp=True)
dev = {"next": out.detach().cpu().numpy()[wd], "grad_state": st.grad.cpu().numpy()[wd], "grad_action": at2.grad.cpu().numpy()[wd]}
print("step 2: status dev", hex(st2[wd]), "ref", hex(r2["status"][0]), {k: float(f"{np.abs(dev[k] - r2[k][0]).max():.2e}") for k in dev})
c2 = world.lcp_cache.cpu().numpy()
print("   device solution", c2[:int(c2[-1, wd]), wd]); print("   oracle solution", r2["lcp"][0][:r2["lcp_len"][0]])
ow.set_lcp_cache(dev_lcp[wd][:dev_len[wd]]); ow.step(s1n[wd], a[wd]); L = ow.last_lcp()
print("   oracle LCP: x", L["x"], "\n   b", L["b"], "\n   lo", L["lo"], "\n   hi", L["hi"], "\n   classes", L["row_class"], "\n   A\n", L["A"])
q = s1n[wd][:md.num_dofs]; fl = md.flat()
print("   limited DOFs at step 2:", [(d, float(q[d]), float(fl["pos_lo"][d]), float(fl["pos_hi"][d])) for d in range(md.num_dofs) if fl["dof_limit_enforced"][d] and (q[d] <= fl["pos_lo"][d] or q[d] >= fl["pos_hi"][d])])
print("   next dev", dev["next"][md.num_dofs:], "\n   next ref", r2["next"][0][md.num_dofs:])
