import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, time
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
np.set_printoptions(linewidth=220, precision=6, suppress=True)
rng = np.random.default_rng(7)
def rel(x, y): return np.abs(x-y).max()/max(np.abs(y).max(), 1e-30)
# regression: no-contact path after the refactor
md = na.atlas("atlas20"); w = na.World(md); ow = OracleWorld(md); n = w.n; B = 64
q = rng.uniform(-0.3, 0.3, (B, n)); q[:, 0] -= 1.5; v = rng.normal(0, 0.3, (B, n)); a = rng.normal(0, 1, (B, n)); s = np.concatenate([q, v], 1); g = rng.normal(0, 1, s.shape)
st = torch.tensor(s, device="cuda", requires_grad=True); at = torch.tensor(a, device="cuda", requires_grad=True)
timestep(w, st, at).backward(torch.tensor(g, device="cuda")); ref = ow.step_batch(s, a, g, threads=8)
print("no-contact regression gstate", rel(st.grad.cpu().numpy(), ref["grad_state"]), "gaction", rel(at.grad.cpu().numpy(), ref["grad_action"]))
for name, noise, vn, an in (("atlas20", 0.0, 0.0, 0.0), ("atlas20", 0.002, 0.001, 0.1), ("atlas33", 0.002, 0.001, 0.1), ("atlas20", 0.02, 0.0, 0.0)):
    md = na.atlas(name, ground=True)
    w = na.World(md); ow = OracleWorld(md); n = w.n
    B = 128
    q = np.zeros((B, n)); q[:, 0] = -np.pi/2; q[:, 4] = -0.01
    q[:, 6:] = rng.normal(0, noise, (B, n-6))
    v = rng.normal(0, vn, (B, n)); a = rng.normal(0, an, (B, n))
    s = np.concatenate([q, v], 1); g = rng.normal(0, 1, s.shape)
    st = torch.tensor(s, device="cuda", requires_grad=True); at = torch.tensor(a, device="cuda", requires_grad=True)
    w.reset_lcp_cache()
    out = timestep(w, st, at)
    status = w.last_status.cpu().numpy().astype(np.uint32)
    out.backward(torch.tensor(g, device="cuda"))
    t0 = time.time(); ref = ow.step_batch(s, a, g, threads=8); t1 = time.time()
    ok = ((status & 0x2) != 0)
    gs, ga = st.grad.cpu().numpy(), at.grad.cpu().numpy()
    es = np.abs(gs - ref["grad_state"]).max(1) / np.abs(ref["grad_state"]).max(); ea = np.abs(ga - ref["grad_action"]).max(1) / np.abs(ref["grad_action"]).max()
    en = np.abs(out.detach().cpu().numpy() - ref["next"]).max(1) / np.abs(ref["next"]).max()
    print(name, noise, "stage0 lanes", ok.sum(), "/", B, "next", en[ok].max(), "gstate", es[ok].max(), "gaction", ea[ok].max(), "oracle s/world", (t1-t0)/B*8)
    worst = np.argmax(np.where(ok, es, 0))
    if es[ok].max() > 1e-6:
        print("  worst lane", worst, "gq diff", (gs - ref["grad_state"])[worst, :n]); print("  gv diff", (gs - ref["grad_state"])[worst, n:]); print("  ref gq", ref["grad_state"][worst, :n])
