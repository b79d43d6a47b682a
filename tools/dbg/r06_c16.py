"""Developer: fwd+bwd time of worlds that FILL 12 / 16 contact slots (cube towers, the table on four feet) on the 48-row build against
the general build (NBL_MIN_VARIANT=2)."""
import sys, os, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import nimblephysics_amd as na
from util import cube_tower_inputs, table_inputs
B = 2048
for name, (md, s, a) in (("tower3 (12 contacts)", cube_tower_inputs(B, 3, 3, max_contacts=16)), ("tower4 (16 contacts)", cube_tower_inputs(B, 4, 4, max_contacts=16)),
                         ("table (16 contacts, rank 6)", table_inputs(B, 5))):
    world = na.World(md, device="cuda:0")
    st = world.to_soa(torch.tensor(s, device="cuda:0")); at = world.to_soa(torch.tensor(a, device="cuda:0"))
    g = torch.randn_like(st)
    def once():
        nxt, saved, status = world.step_soa(st, at)
        world.backward_soa(saved, g)
        return status
    for _ in range(3): status = once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): once()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    stv = status.cpu().numpy()
    print(f"{name:30s} slots {world._L.nbl_model_max_contacts(world._h)}: {dt * 1e3:.3f} ms / step of {B} worlds = {B / dt / 1e6:.3f} M/s; cascade share {(stv & 2 != 0).mean():.2f}")
