"""Debug: per-lane comparison of the wave-cooperative dense kernels (NBL_COOP=1) with the one-world-per-lane ones."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from util import contact_inputs

def run(coop, md, s, a, g):
    os.environ["NBL_COOP"] = "1" if coop else "0"
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True)
    at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy()
    out.backward(torch.tensor(g, device="cuda:0"))
    return out.detach().cpu().numpy(), st.grad.cpu().numpy(), at.grad.cpu().numpy(), status

md, s, a = contact_inputs("atlas20", 1024, 13, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
g = np.random.default_rng(5).normal(0, 1, s.shape)
o1 = run(True, md, s, a, g)
o0 = run(False, md, s, a, g)
for name, x, y in (("next", o1[0], o0[0]), ("gs", o1[1], o0[1]), ("ga", o1[2], o0[2])):
    e = np.abs(x - y).max(1) / np.maximum(np.abs(y).max(1), 1e-12)
    bad = np.where(e > 1e-7)[0]
    print(name, "max", e.max(), "bad lanes", len(bad), bad[:10], [hex(o1[3][i]) for i in bad[:10]], [hex(o0[3][i]) for i in bad[:10]], e[bad[:10]])
print("status equal", np.array_equal(o1[3], o0[3]))

# ---- dump the lws rows written by the dense adjoint kernel for one bad lane under both modes ----
def run2(coop):
    os.environ["NBL_COOP"] = "1" if coop else "0"
    world = na.World(md, device="cuda:0")
    B = s.shape[0]
    st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
    nxt, saved, status = world.step_soa(st, at)
    gs, ga = world.backward_soa(saved, world.to_soa(torch.tensor(g, device="cuda:0")))
    torch.cuda.synchronize()
    ws = world._workspace(B).view(torch.float64) if world._workspace(B).dtype != torch.float64 else world._workspace(B)
    return ws.cpu().numpy(), saved.view(torch.float64).cpu().numpy(), B
w1, sv1, B = run2(True)
w0, sv0, _ = run2(False)
LBT = 1993
lw1 = w1[: (w1.size // B) * B].reshape(-1, B); lw0 = w0[: (w0.size // B) * B].reshape(-1, B)
off = lw1.shape[0] - LBT - 1
# find offset of lws: rows after nb*288
for cand in range(lw1.shape[0] - LBT, -1, -288):
    pass
nbod = (lw1.shape[0] - LBT) // 288
off = nbod * 288
print("rows", lw1.shape, "nb", nbod)
e = np.abs(o1[1] - o0[1]).max(1) / np.maximum(np.abs(o0[1]).max(1), 1e-12)
lane = int(np.where(e > 1e-7)[0][0])
names = {"LAM1": (1440, 20), "GVP": (1480, 20), "S": (1560, 120), "P": (1680, 120), "COEF": (1800, 192), "FLAG": (1992, 1)}
for k, (r0, cnt) in names.items():
    x1 = lw1[off + r0: off + r0 + cnt, lane]; x0 = lw0[off + r0: off + r0 + cnt, lane]
    print(k, "maxdiff", np.abs(x1 - x0).max(), "scale", np.abs(x0).max())
    if k == "COEF":
        d = np.abs(x1 - x0).reshape(24, 8); print(np.round(d / max(np.abs(x0).max(), 1e-30), 3))
n = 20
svr1 = sv1[: 351 * B].reshape(351, B)
print("lane", lane, "nc", svr1[100, lane], "cls", svr1[325:349, lane], "x", np.round(svr1[277:301, lane], 4), "pflag", svr1[350, lane])
