import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, time
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
from collections import Counter
rng = np.random.default_rng(21)
for name, noise, vn, an, B in (("atlas20", 0.02, 0.01, 0.0, 1024), ("atlas20", 0.02, 0.0, 1.0, 1024), ("atlas33", 0.02, 0.01, 0.5, 512)):
    md = na.atlas(name, ground=True)
    w = na.World(md); ow = OracleWorld(md); n = w.n
    q = np.zeros((B, n)); q[:, 0] = -np.pi/2; q[:, 4] = -0.01
    q[:, 6:] = rng.normal(0, noise, (B, n-6))
    v = rng.normal(0, vn, (B, n)); a = rng.normal(0, an, (B, n))
    s = np.concatenate([q, v], 1); g = rng.normal(0, 1, s.shape)
    st = torch.tensor(s, device="cuda", requires_grad=True); at = torch.tensor(a, device="cuda", requires_grad=True)
    w.reset_lcp_cache()
    out = timestep(w, st, at)
    status = w.last_status.cpu().numpy().astype(np.uint32)
    out.backward(torch.tensor(g, device="cuda"))
    ref = ow.step_batch(s, a, g, threads=8)
    ost = ref["status"] & 0x3ff
    same = (status & 0x1ff) == (ost & 0x1ff)
    en = np.abs(out.detach().cpu().numpy() - ref["next"]).max(1) / np.abs(ref["next"]).max()
    es = np.abs(st.grad.cpu().numpy() - ref["grad_state"]).max(1) / np.abs(ref["grad_state"]).max()
    good = en < 1e-7
    print(name, noise, vn, an, "gpu", dict(Counter(hex(x & 0x1ff) for x in status)))
    print("     oracle", dict(Counter(hex(x & 0x1ff) for x in ost)))
    print("     status agree", same.mean(), " next<1e-7:", good.mean(), " among status-agree lanes next max", en[same].max(), "grad max", es[same & good].max())
    bad = np.where(~good)[0][:5]
    for i in bad: print("       lane", i, hex(status[i]), hex(ost[i]), en[i])
    gbad = np.where((es > 1e-6) & good)[0]
    print("     grad-bad lanes", len(gbad), [(int(i), hex(status[i]), float(es[i])) for i in gbad[:8]])
    for i in gbad[:3]:
        o1 = OracleWorld(md); o1.step(s[i], a[i]); l = o1.last_lcp(); print("       lane", i, "classes", l["row_class"], "types", o1.last_contacts()[:, 7])
