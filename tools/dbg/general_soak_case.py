"""developer aid: one world of a soak seed on the 24- / 48-row build and on the general build, next to the oracle.
   usage: general_soak_case.py <seed> <world> [stress mode] [variant]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import copy
import numpy as np
import torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from oracle import OracleWorld
import soak_parity, soak_stress

seed, wd = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else ""
variant = sys.argv[4] if len(sys.argv) > 4 else "balls"
np.set_printoptions(linewidth=220, precision=6)


def case(slots):
    if mode:
        md, s, a, g = soak_parity.make_case(seed, 256, variant == "big", variant == "multi", variant == "balls", False, slots)
        md, s, a, g = soak_stress.mutator(mode, slots)(seed, md, s, a, g)
        if slots is not None:
            md.max_contacts = slots
    else:
        md, s, a, g = soak_parity.make_case(seed, 256, False, False, False, False, slots)
    return md, s, a, g


res = {}
for slots in (None, 64):
    md, s, a, g = case(slots)
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0", requires_grad=True); at = torch.tensor(a, device="cuda:0", requires_grad=True)
    out = timestep(world, st, at)
    status = world.last_status.cpu().numpy().astype(np.uint32)
    cache = world.lcp_cache.cpu().numpy()
    out.backward(torch.tensor(g, device="cuda:0"))
    ow = OracleWorld(md)
    ref = ow.step_batch(s, a, g, threads=8)
    dev = {"next": out.detach().cpu().numpy(), "grad_state": st.grad.cpu().numpy(), "grad_action": at.grad.cpu().numpy()}
    print(f"== slots {slots}: build max contacts {world._L.nbl_model_max_contacts(world._h)}, status dev {status[wd]:#x} ref {ref['status'][wd]:#x}; n {md.num_dofs}")
    for k in dev:
        sc = np.abs(ref[k]).max()
        print(f"   {k}: |dev - ref| / scale = {np.abs(dev[k][wd] - ref[k][wd]).max() / sc:.3e}   |dev| {np.abs(dev[k][wd]).max():.3e} |ref| {np.abs(ref[k][wd]).max():.3e}")
    m = int(cache[-1, wd])
    print("   device LCP rows", m, "x:", cache[:m, wd])
    ow.reset_lcp_cache(); ow.step(s[wd], a[wd]); L = ow.last_lcp()
    print("   oracle LCP rows", len(L["b"]), "x:", L["x"], "\n   b:", L["b"], "\n   findex:", L["findex"])
    if L["A"].size:
        sv = np.linalg.svd(L["A"], compute_uv=False)
        print("   singular values of the oracle's A:", sv)
    res[slots] = dev
for k in res[None]:
    print(f"general vs small build, {k}: {np.abs(res[None][k][wd] - res[64][k][wd]).max():.3e}")
