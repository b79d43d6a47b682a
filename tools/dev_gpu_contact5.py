import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from collections import Counter
rng = np.random.default_rng(9)
md = na.box_stack(); w = na.World(md); ow = OracleWorld(md); n = w.n
B = 256; pen = 5e-4
q = np.zeros((B, 12))
q[:, 1] = rng.uniform(-np.pi, np.pi, B); q[:, 3] = rng.uniform(0.9, 0.99, B) * rng.choice([-1, 1], B); q[:, 4] = 0.1 - pen; q[:, 5] = rng.uniform(-0.5, 0.5, B)
q[:, 10] = 5.0
v = np.zeros((B, 12)); v[:, 0:6] = rng.normal(0, 0.001, (B, 6)); a = np.zeros((B, 12))
s = np.concatenate([q, v], 1); g = rng.normal(0, 1, s.shape)
st = torch.tensor(s, device="cuda", requires_grad=True); at = torch.tensor(a, device="cuda", requires_grad=True)
out = timestep(w, st, at); status = w.last_status.cpu().numpy().astype(np.uint32)
out.backward(torch.tensor(g, device="cuda"))
ref = ow.step_batch(s, a, g, threads=8)
en = np.abs(out.detach().cpu().numpy() - ref["next"]).max(1) / np.abs(ref["next"]).max()
es = np.abs(st.grad.cpu().numpy() - ref["grad_state"]).max(1) / np.abs(ref["grad_state"]).max()
print("gpu", dict(Counter(hex(x) for x in status))); print("ora", dict(Counter(hex(x) for x in ref["status"])))
ok = en < 1e-7
print("next agree", ok.mean(), "grad err max over agreeing lanes", es[ok].max(), "stage0 lanes grad max", es[ok & ((status & 2) != 0)].max())
bad = np.where(ok & (es > 1e-6))[0]
print("bad lanes", len(bad))
for i in bad[:4]:
    o = OracleWorld(md); o.step(s[i], a[i]); print("  lane", i, hex(status[i]), es[i], "types", o.last_contacts()[:, 7].astype(int), "cls", o.last_lcp()["row_class"])
