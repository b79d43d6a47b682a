"""Developer: cycles per phase of the general build's solve kernel (k_contact_solve_gen) on the metric worlds.  Needs
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DNBL_MAXC=64 -DNBL_GEN_TIMING nimblephysics_amd/csrc/nimble_amd.hip -o tools/dbg/libnimble_amd_gentiming.so
usage (GPU box): python tools/gen_timing.py [worlds] [joint noise]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nimblephysics_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "dbg", "libnimble_amd_gentiming.so")
import nimblephysics_amd as na
from util import contact_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
jn = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
md, s, a = contact_inputs("atlas20", B, 1000, joint_noise=jn, vel_noise=jn / 2, action_noise=0.1)
md.max_contacts = 24
world = na.World(md, device="cuda:0")
world.set_slices(1)
st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
L = _lib.lib()
buf = (ctypes.c_ulonglong * 32)()
world.step_soa(st, at); torch.cuda.synchronize()          # warm-up
L.nbl_debug_gen_stats(buf, 1)
world.reset_lcp_cache()                                   # a COLD LCP start, like every step of bench.py's timed region
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); nxt, saved, status = world.step_soa(st, at); t1.record(); torch.cuda.synchronize()
L.nbl_debug_gen_stats(buf, 0)
g = list(buf)
nw, ncas = max(g[10], 1), max(g[11], 1)
print(f"{B} worlds, forward step {t0.elapsed_time(t1):.3f} ms; {g[10]} worlds in k_contact_solve_gen, {g[11]} cascades")
for k, nm, den in ((0, "rows, groups, final classification", nw), (1, "stage 0 (guess, classification, standardisation)", nw), (12, "cascade (all of it)", ncas),
                   (4, "  stage 1: load + reduce", ncas), (5, "  stage 1: Dantzig", ncas), (6, "  stage 1: map out + validity", ncas),
                   (7, "  stage 2: CFM + reduce + PGS + validity", ncas), (8, "  stage 3: frictionless PGS (+ NaN checks)", ncas),
                   (9, "  standardisation loop of the chosen solution", ncas),
                   (13, "  (Gauss-Seidel, both stages: set-up - scaling, transposed matrix, residuals)", ncas), (14, "  (Gauss-Seidel, both stages: the sweeps)", ncas), (2, "the record's Q^+ when the last one is not it", nw), (3, "outputs", nw)):
    print(f"  {nm:56s} {g[k] / den:12.0f} cycles per {'world' if den == nw else 'cascade'}")

# finer stamps (round 6): stage 0's guess, the standardisation loop (stage 0's and the cascades' together), the pseudo-inverse
print(f"  stage 0: guess rows + matrix {g[16] / nw:10.0f}  its pseudo-inverse {g[17] / nw:10.0f}  apply + X0 {g[18] / nw:10.0f}   cycles per world")
it, pv = max(g[24], 1), max(g[25], 1)
print(f"  standardisation loops: {g[24]} iterations ({g[24] / nw:.2f} per world), {g[25]} with a factorisation; per iteration: classify {g[19] / it:8.0f}  "
      f"build Q {g[20] / pv:8.0f} (per factorisation)  pseudo-inverse {g[21] / pv:8.0f} (per factorisation)  apply + new x {g[22] / it:8.0f}  validity {g[23] / it:8.0f}")
pc = max(g[29], 1)
print(f"  pseudo-inverse: {g[29]} calls ({g[29] / nw:.2f} per world), {g[30]} rank-deficient; per call: pivoted Householder QR {g[26] / pc:8.0f}  R1^-1 solves {g[27] / pc:8.0f}  "
      f"completion / output {g[28] / pc:8.0f}")
