import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from util import folding_arm
import test_gpu_self_collision as t
a0 = folding_arm(True, "sphere")
bodies, boxes = [], []
for k, dz in enumerate((0.0, 0.076)):
    for b in copy.deepcopy(a0.bodies):
        b.name += f"_{k}"; b.joint_name += f"_{k}"; b.parent = b.parent if b.parent < 0 else b.parent + 3 * k
        if b.parent < 0: b.T_pj = na.make_transform((0, 0, dz))
        b.skeleton = k; bodies.append(b)
    for bx in copy.deepcopy(a0.boxes):
        bx.body += 3 * k; boxes.append(bx)
md = na.ModelDescription("two_arms", bodies, boxes, gravity=(0, -9.81, 0), max_contacts=8)
rng = np.random.default_rng(5); B = 256
s1, a1 = t._states(B, 6, 1.895, 1.94); s2, a2 = t._states(B, 7, 1.895, 1.94)
s2[:, :3] = s1[:, :3] + rng.normal(0, 0.015, (B, 3))
s = np.concatenate([s1[:, :3], s2[:, :3], s1[:, 3:], s2[:, 3:]], 1); a = np.concatenate([a1, a2], 1)
g = np.random.default_rng(8).normal(0, 1, s.shape)
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
out = timestep(world, st, at); out.backward(torch.tensor(g, device="cuda:0"))
ref = ow.step_batch(s, a, g, threads=8)
status = world.last_status.cpu().numpy().astype(np.uint32)
np.set_printoptions(linewidth=200, precision=3)
for k, dv in (("next", out.detach().cpu().numpy()), ("grad_state", st.grad.cpu().numpy()), ("grad_action", at.grad.cpu().numpy())):
    e = np.abs(dv - ref[k]).max(1) / np.abs(ref[k]).max()
    print(k, "bad", (e > 1e-7).sum(), "max", e.max())
e = np.abs(out.detach().cpu().numpy() - ref["next"]).max(1)
eg = np.abs(st.grad.cpu().numpy() - ref['grad_state']).max(1) / np.abs(ref['grad_state']).max()
print('bad grad worlds', np.where(eg > 1e-7)[0][:20], 'status', [hex(x) for x in status[np.where(eg > 1e-7)[0][:20]]])
for wd in np.where(eg > 1e-7)[0][:5]:
    ow.reset_lcp_cache(); ow.step(s[wd], a[wd]); c = ow.last_contacts()
    print(wd, hex(status[wd]), hex(ref["status"][wd]), "types", c[:, 7], "boxes", c[:, 8:10].astype(int).tolist(), "depth", c[:, 6])
    print("  dnext", out.detach().cpu().numpy()[wd] - ref["next"][wd])
