"""The mirrored `World` API on the GPU (World.cpp:221-254, 2016-2135, 1821-1824): step() on models with colliders, action-space
changes, mass updates."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_world_step_on_a_contact_model_equals_timestep():
    """World::step (no gradient bookkeeping) on a model WITH colliders: the library needs the saved record as contact scratch
    even when no backward pass is wanted; the result must equal timestep()'s and the oracle's."""
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    from util import contact_inputs, rel_err
    md, s, a = contact_inputs("atlas20", 256, 41)
    world = na.World(md, device="cuda:0")
    st = torch.tensor(s, device="cuda:0"); at = torch.tensor(a, device="cuda:0")
    ref = timestep(world, st, at)
    world.reset_lcp_cache()
    world.setState(st); world.setAction(at)
    world.step()
    got = world.getState()
    assert torch.equal(got, ref)
    assert (world.last_status.cpu().numpy() & 0x1).all()
    assert rel_err(got.cpu().numpy(), OracleWorld(md).step_batch(s, a, threads=4)["next"]) < 1e-7
    # a second step continues from the first (warm-started LCP), like chained timestep() calls
    world.step()
    w2 = na.World(md, device="cuda:0")
    chained = timestep(w2, timestep(w2, st, at), at)
    assert torch.equal(world.getState(), chained)


def test_set_action_space_recreates_the_handle_and_keeps_working():
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    from util import cfg_inputs, rel_err
    md, s, a = cfg_inputs("cartpole", 64, 42)
    world = na.World(md, device="cuda")          # a bare "cuda": resolved to the current device
    assert world.device.index == torch.cuda.current_device()
    for _ in range(20):                          # every call used to leak the previous handle (buffers, streams, events)
        world.setActionSpace([0])
        world.setActionSpace([0, 1])
    world.removeDofFromActionSpace(1)
    assert world.getActionSize() == 1 and world.getActionSpace() == [0]
    st = torch.tensor(s, device="cuda:0"); at = torch.tensor(a[:, :1], device="cuda:0")
    out = timestep(world, st, at)
    md2 = na.cartpole(); md2.set_action_space([0])
    ref = OracleWorld(md2).step_batch(s, a[:, :1], threads=2)["next"]
    assert rel_err(out.cpu().numpy(), ref) < 1e-9


def test_set_masses_uploads_only_changes_and_matches_the_oracle():
    import torch
    import nimblephysics_amd as na
    from nimblephysics_amd.timestep import timestep
    from oracle import OracleWorld
    from util import cfg_inputs, rel_err
    md, s, a = cfg_inputs("atlas20", 32, 43)
    world = na.World(md, device="cuda:0")
    world.tuneMass(3); world.tuneMass(7)
    m0 = world.getMasses().numpy().copy()
    calls = []
    orig = world._L.nbl_set_body_inertias
    world._L.nbl_set_body_inertias = lambda *args: (calls.append(int(args[1])), orig(*args))[1]
    try:
        world.setMasses(m0)                      # unchanged: nothing is uploaded
        assert calls == []
        world.setMasses(m0 * np.array([1.5, 1.0]))
        assert calls == [1]                      # one body changed -> one upload of one body
    finally:
        world._L.nbl_set_body_inertias = orig
    st = torch.tensor(s, device="cuda:0"); at = torch.tensor(a, device="cuda:0")
    out = timestep(world, st, at)
    ref = OracleWorld(world.description).step_batch(s, a, threads=2)["next"]
    assert rel_err(out.cpu().numpy(), ref) < 1e-9
