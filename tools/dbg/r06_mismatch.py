import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import nimblephysics_amd as na
import soak_parity, soak_stress
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
seed, wd, B = int(sys.argv[1]), int(sys.argv[2]), 256
md, s, a, g = soak_parity.make_case(seed, B, balls=True)
md, s, a, g = soak_stress.mutator("mix")(seed, md, s, a, g)
print("model:", md.name, "dofs", md.num_dofs, "bodies", [(b.name, b.joint_type) for b in md.bodies][:12], "max_contacts", md.max_contacts, "dt", md.dt)
print("limits enforced:", getattr(md, "dof_limit_enforced", None))
world = na.World(md, device="cuda:0")
ow = OracleWorld(md); ow.set_lcp_cache_slots(True)
at = torch.tensor(a, device="cuda:0")
with torch.no_grad():
    s1 = timestep(world, torch.tensor(s, device="cuda:0"), at)
st1 = world.last_status.cpu().numpy().astype(np.uint32)
r1 = ow.step_batch(s, a, None, threads=8, want_lcp=True)
s1n = s1.cpu().numpy()
print("step 1: status dev", hex(int(st1[wd])), "ref", hex(int(r1["status"][wd])), "err next", np.abs(s1n[wd] - r1["next"][wd]).max())
cache = world.lcp_cache.clone().cpu().numpy()
st = s1.clone().requires_grad_(True); at2 = at.clone().requires_grad_(True)
out = timestep(world, st, at2)
st2 = world.last_status.cpu().numpy().astype(np.uint32)
out.backward(torch.tensor(g, device="cuda:0"))
dev_lcp = np.ascontiguousarray(cache[:r1["lcp"].shape[1]].T); dev_len = cache[-1].astype(np.int32)
r2 = ow.step_batch(s1n, a, g, threads=8, lcp_in=dev_lcp, lcp_len_in=dev_len)
dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at2.grad.cpu().numpy()}
n = md.num_dofs
for k in dev:
    sc = max(np.abs(r2[k]).max(), 1e-30)
    d = np.abs(dev[k][wd] - r2[k][wd])
    print(k, "max err", d.max() / sc, "at", int(d.argmax()), "scale", sc, "dev", dev[k][wd][d.argmax()], "ref", r2[k][wd][d.argmax()])
print("status step 2 dev", hex(int(st2[wd])), "ref", hex(int(r2["status"][wd])))
print("near log-map singularity:", soak_parity.near_log_map_singularity(md, r2["next"][wd]))
print("q:", np.round(s1n[wd][:n], 4)); print("v:", np.round(s1n[wd][n:], 3))
d = np.abs(dev["grad_state"][wd] - r2["grad_state"][wd]); print("grad_state err by entry:", np.round(d / max(np.abs(r2["grad_state"]).max(), 1e-30), 8))
# who is right?  central differences of the forward step (device and oracle) in state entry e
e = int(np.argmax(d))
for h in (1e-4, 1e-5, 1e-6):
    sp = s1n.copy(); sm = s1n.copy(); sp[wd, e] += h; sm[wd, e] -= h
    world.reset_lcp_cache()
    with torch.no_grad():
        fp = timestep(world, torch.tensor(sp, device="cuda:0"), at).cpu().numpy()[wd]
        world.reset_lcp_cache()
        fm = timestep(world, torch.tensor(sm, device="cuda:0"), at).cpu().numpy()[wd]
    op = ow.step_batch(sp[wd:wd + 1], a[wd:wd + 1], None, threads=1)["next"][0]; om = ow.step_batch(sm[wd:wd + 1], a[wd:wd + 1], None, threads=1)["next"][0]
    print(f"h {h:g}: g . d next / d s[{e}] by central differences: device forward {np.dot(g[wd], (fp - fm) / (2 * h)):.10f}  oracle forward {np.dot(g[wd], (op - om) / (2 * h)):.10f}   (device backward {dev['grad_state'][wd][e]:.10f}, oracle backward {r2['grad_state'][wd][e]:.10f})")
col = (fp - fm) / (2 * h)
print("column d next / d s[e]:", np.round(col, 6))
print("g:", np.round(g[wd], 4))
