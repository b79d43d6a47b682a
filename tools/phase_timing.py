"""Developer tool: where does one wavefront of k_step_forward_coop spend its cycles?
Builds a -DNBL_PHASE_TIMING copy of the library into gpurun_out/, runs a few steps and prints the cycle stamps
(clock64 = s_memtime, 100 MHz constant clock on gfx950: 10 ns per tick) between the phase boundaries."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
out = os.path.join(ROOT, "tools", "libnimble_amd_phase.so")     # git-ignored (*.so); build it here, it travels with gpurun
if "--build" in sys.argv or not os.path.exists(out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DNBL_PHASE_TIMING",
                           os.path.join(ROOT, "nimblephysics_amd/csrc/nimble_amd.hip"), "-o", out, "-w"])
    if "--build" in sys.argv:
        sys.exit(0)
import torch
import nimblephysics_amd as na
from nimblephysics_amd import _lib
_lib.LIB_PATH = out
from util import contact_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
md, s, a = contact_inputs("atlas20", B, 1000)
w = na.World(md, device="cuda:0")
x = w.to_soa(torch.tensor(s, device="cuda:0")); u = w.to_soa(torch.tensor(a, device="cuda:0"))
L = _lib.lib()
names = {0: "kernel entry", 1: "setup (model -> LDS, barrier)", 2: "joint transforms (q, v loads)", 3: "sweep 1 (TW, V down)", 4: "own inertia / bias",
         5: "sweep 2 (AI, bias up)", 6: "sweep 3 (acc down)", 7: "kept slots", 8: "integrate + saved rows", 9: "pre-contact twists", 10: "store tree"}
bnames = {12: "recompute: setup", 13: "recompute: flag loads", 14: "recompute: load tree", 15: "recompute: world body", 16: "recompute: minv sweeps",
          17: "recompute: store lambda1", 21: "final: setup", 22: "final: load tree", 23: "final: world body", 24: "final: minv sweeps",
          25: "final: reverse, local part", 26: "final: reverse, level loop", 27: "final: epilogue (VJPs, stores)"}
g = torch.randn_like(x)
for it in range(3):
    nxt, saved, _ = w.step_soa(x, u)
    w.backward_soa(saved, g)
    buf = (C.c_ulonglong * 64)()
    L.nbl_debug_phase_stamps.argtypes = [C.c_void_p]
    L.nbl_debug_phase_stamps(buf)
    t = list(buf)
    print(f"-- step {it}: total {t[10] - t[0]} ticks")
    for k in range(1, 11):
        print(f"   {names[k]:34s} {t[k] - t[k - 1]:8d}")
    print(f"   recompute total {t[17] - t[11]}, final total {t[27] - t[20]}")
    rn = {33: "rows: prologue (staging, lane = body)", 34: "rows: row wrench + sync", 35: "rows: b, A_c column, zero acc", 36: "rows: leaf->root chain",
          37: "rows: root->leaf all bodies", 38: "rows: row of A"}
    print(f"   rows total {t[38] - t[32]}")
    for k in sorted(rn):
        print(f"   {rn[k]:38s} {t[k] - t[k - 1]:8d}")
    sn = {41: "solve: nc + load row (mu, b, |A col|)", 42: "solve: warm-start check", 43: "solve: guess mask + A column", 44: "solve: pinv (QR + COD)",
          45: "solve: pinv apply", 46: "solve: standardise loop", 47: "solve: outputs"}
    print(f"   solve total {t[47] - t[40]}")
    for k in sorted(sn):
        print(f"   {sn[k]:38s} {t[k] - t[k - 1]:8d}")
    bb = {49: "bwd_b: staging (TW, contact bodies)", 50: "bwd_b: 1a twist fields + prefix", 51: "bwd_b: 2 row constants (loads, terms)",
          52: "bwd_b: 2 sides A/B scatter", 53: "bwd_b: 1b local wrenches", 54: "bwd_b: 3 subtree sums", 55: "bwd_b: 4 projection + stores"}
    print(f"   bwd_b total {t[55] - t[48]}")
    for k in sorted(bb):
        print(f"   {bb[k]:38s} {t[k] - t[k - 1]:8d}")
    dn_ = {56: "detect: joint transforms of the collider chains (all threads)", 57: "detect: narrow phase of the lane's pair (thread 0)",
           58: "detect: barrier (the slowest lane of the workgroup)", 59: "detect: accept parked contacts", 60: "detect: limits, count, status"}
    print(f"   detect workgroup 0: kernel entry -> end {t[60] - t[18]} (the last tree workgroup: {t[10] - t[0]}); staging of the body and collider constants {t[19] - t[18]}")
    print(f"   box-box of thread 0: start -> context {int(t[29]) - int(t[56])}, -> chain mask loaded {int(t[30]) - int(t[29])}, first link {int(t[31]) - int(t[30])}, rest of the chain + collider offset {int(t[28]) - int(t[31])}, -> entry {int(t[61]) - int(t[28])}, 15 axes {int(t[62]) - int(t[61])}, face set-up {int(t[39]) - int(t[62])}, clip {int(t[63]) - int(t[39])}, points + accept {int(t[57]) - int(t[63])}")
    for k in sorted(dn_):
        print(f"   {dn_[k]:38s} {t[k] - (t[19] if k == 56 else t[k - 1]):8d}")
    for k in sorted(bnames):
        print(f"   {bnames[k]:34s} {t[k] - t[k - 1]:8d}")
