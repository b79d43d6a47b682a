import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, time
import nimblephysics_amd as na
from oracle import OracleWorld
from collections import Counter
rng = np.random.default_rng(5)
for name, noise, vn in (("atlas20", 0.002, 0.001), ("atlas20", 0.02, 0.0), ("atlas33", 0.002, 0.001)):
    md = na.atlas(name, ground=True)
    w = na.World(md); ow = OracleWorld(md); n = w.n
    B = 256
    q = np.zeros((B, n)); q[:, 0] = -np.pi/2; q[:, 4] = -0.01
    q[:, 6:] = rng.normal(0, noise, (B, n-6))
    v = rng.normal(0, vn, (B, n)); a = rng.normal(0, 1, (B, n))
    s = np.concatenate([q, v], 1)
    st = w.to_soa(torch.tensor(s, device="cuda")); at = w.to_soa(torch.tensor(a, device="cuda"))
    nxt, saved, status = w.step_soa(st, at)
    torch.cuda.synchronize()
    out = w.from_soa(nxt).cpu().numpy(); stg = status.cpu().numpy().astype(np.uint32)
    ref = ow.step_batch(s, a, threads=8)
    ok = (stg & 0x20) == 0
    err = np.abs(out - ref["next"]).max(1) / np.abs(ref["next"]).max()
    print(name, noise, "gpu status", dict(Counter(hex(x) for x in stg)), "oracle", dict(Counter(hex(x) for x in ref["status"])))
    print("   max rel err (resolved lanes)", err[ok].max() if ok.any() else None, " unresolved:", (~ok).sum(), "err on unresolved", err[~ok].max() if (~ok).any() else None)
    # resolved lanes should be exactly the oracle's stage-0 lanes
    o0 = (ref["status"] & 0x2) != 0
    print("   stage0 agreement:", (ok == o0).mean())
