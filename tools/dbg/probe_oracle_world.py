"""Debug: how stable is the ORACLE's outcome of one soak world under k-ulp perturbations of its inputs?
usage: python tools/dbg/probe_oracle_world.py <seed> <world> [big|multi|balls]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("NBL_LIB_PATH", "")
import importlib
import types
# soak_parity imports torch + the device package at module level; only make_case is needed here
src = open(os.path.join(ROOT, "tools", "soak_parity.py")).read().replace("import torch  # noqa: E402", "").replace(
    "from nimblephysics_amd.timestep import timestep  # noqa: E402", "")
mod = types.ModuleType("soak_cpu"); mod.__file__ = os.path.join(ROOT, "tools", "soak_parity.py"); exec(compile(src, "soak_cpu", "exec"), mod.__dict__)
from oracle import OracleWorld
seed, wd = int(sys.argv[1]), int(sys.argv[2]); mode = sys.argv[3] if len(sys.argv) > 3 else ""
md, s, a, g = mod.make_case(seed, 256, big=mode == "big", multi=mode == "multi", balls=mode == "balls")
ow = OracleWorld(md)
base = ow.step_batch(s[wd][None], a[wd][None], g[wd][None])
print("base status", hex(int(base["status"][0])))
rng = np.random.default_rng(1)
for ulps in (1, 4, 16, 64, 256, 4096):
    sp = s[wd][None] * (1.0 + rng.choice([-1.0, 0.0, 1.0], (128, s.shape[1])) * ulps * 2.220446049250313e-16)
    r = ow.step_batch(sp, np.repeat(a[wd][None], 128, 0), np.repeat(g[wd][None], 128, 0), threads=8)
    st, cnt = np.unique(r["status"], return_counts=True)
    spread = max(np.abs(r[k] - base[k]).max() / max(np.abs(base[k]).max(), 1e-30) for k in ("next", "grad_state", "grad_action"))
    print(f"{ulps:5d} ulps: statuses {dict(zip([hex(int(x)) for x in st], cnt.tolist()))} spread {spread:.2e}")
