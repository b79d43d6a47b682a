"""GPU: T chained steps of one mixed-soak model on the device and in the oracle: where do non-finite states appear?
usage: python tools/dbg/rollout_nan_dbg.py <seed> <variant> [T]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
import soak_parity, soak_stress
seed = int(sys.argv[1]); variant = sys.argv[2]; T = int(sys.argv[3]) if len(sys.argv) > 3 else 5; B = 64
pick = np.random.default_rng(seed + 77).random(len(soak_stress.MIX_ORDER)) < 0.5
print("parts", [m for m, p in zip(soak_stress.MIX_ORDER, pick) if p])
md, s, a, g = soak_parity.make_case(seed, B, variant == "big", variant == "multi", variant == "balls", False)
md, s, a, g = soak_stress.mutator("mix")(seed, md, s, a, g)
print("dt", md.dt, "nb", len(md.bodies), "n", md.num_dofs, "limits", [i for i, b in enumerate(md.bodies) if b.limit_enforced], "masses", [round(b.mass, 4) for b in md.bodies])
world = na.World(md, device="cuda:0"); ow = OracleWorld(md)
x = torch.tensor(s, device="cuda:0"); at = torch.tensor(a, device="cuda:0"); xo = s.copy()
lc = None; ll = None
for t in range(T):
    with torch.no_grad():
        x = timestep(world, x, at)
    st = world.last_status.cpu().numpy().astype(np.uint32)
    r = ow.step_batch(xo, a, None, threads=8, lcp_in=lc, lcp_len_in=ll, want_lcp=True)
    lc, ll = r["lcp"], r["lcp_len"]
    xo = r["next"]
    xd = x.cpu().numpy()
    fin_d = np.isfinite(xd).all(1); fin_o = np.isfinite(xo).all(1)
    both = fin_d & fin_o
    err = np.abs(xd[both] - xo[both]).max() / max(np.abs(xo[both]).max(), 1e-30) if both.any() else float("nan")
    print(f"step {t}: finite dev {fin_d.sum()} oracle {fin_o.sum()}; max |v| dev {np.nanmax(np.abs(xd[:, md.num_dofs:])):.3g} oracle {np.nanmax(np.abs(xo[:, md.num_dofs:])):.3g}; "
          f"rel err on finite {err:.2e}; status bits dev {hex(int(np.bitwise_or.reduce(st)))} oracle {hex(int(np.bitwise_or.reduce(r['status'])))}; nan-flag dev {(st & 0x40 != 0).sum()} oracle {(r['status'] & 0x40 != 0).sum()}")
