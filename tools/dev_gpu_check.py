import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, time
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
torch.manual_seed(0)
for mk in (na.single_pendulum, na.cartpole, lambda: na.atlas("atlas33"), lambda: na.atlas("atlas20")):
    md = mk()
    w = na.World(md)
    ow = OracleWorld(md)
    n, k = w.n, w.k
    B = 130
    rng = np.random.default_rng(1)
    q = rng.uniform(-0.5, 0.5, (B, n)); v = rng.normal(0, 0.5, (B, n)); a = rng.normal(0, 1, (B, k))
    if n >= 6: q[:, 0] -= 1.5
    s = np.concatenate([q, v], 1)
    g = rng.normal(0, 1, (B, 2*n))
    st = torch.tensor(s, device="cuda", requires_grad=True); at = torch.tensor(a, device="cuda", requires_grad=True)
    out = timestep(w, st, at)
    out.backward(torch.tensor(g, device="cuda"))
    ref = ow.step_batch(s, a, g, threads=8)
    def rel(x, y): return np.abs(x-y).max()/max(np.abs(y).max(), 1e-30)
    print(md.name, "next", rel(out.detach().cpu().numpy(), ref["next"]), "gstate", rel(st.grad.cpu().numpy(), ref["grad_state"]), "gaction", rel(at.grad.cpu().numpy(), ref["grad_action"]), flush=True)
