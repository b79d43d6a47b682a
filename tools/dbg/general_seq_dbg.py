"""developer aid: the three-groups scene AFTER other worlds in one process (stale allocator blocks), NBL_DEBUG_SYNC=1 names the kernels"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import nimblephysics_amd as na
from util import cube_tower_inputs
import test_gpu_general as tg

def run(md, s, a, tag):
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0"); at = torch.tensor(a, device="cuda:0")
    print("==", tag, "forward", flush=True)
    nxt, sv, status = world.step_soa(world.to_soa(st), world.to_soa(at), want_saved=True)
    torch.cuda.synchronize()
    print("==", tag, "backward", flush=True)
    g = np.random.default_rng(2).normal(0, 1, s.shape)
    gs, ga = world.backward_soa(sv, world.to_soa(torch.tensor(g, device="cuda:0")))
    torch.cuda.synchronize()
    print("==", tag, "done", flush=True)

junk = [torch.full((30000017,), float("nan"), device="cuda:0", dtype=torch.float64) for _ in range(4)]
del junk
md, s, a = cube_tower_inputs(24, 17, 10, max_contacts=48)
run(md, s, a, "tower10")
md, s, a = tg.three_groups_scene(48)
run(md, s, a, "groups")
