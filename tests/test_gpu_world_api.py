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


def test_enforcing_joint_limits_on_a_live_collider_less_world():
    """World.setPositionLimitEnforced on a model WITHOUT colliders: the handle it had carries no LCP at all (0 rows); the new one does, and
    everything sized by the handle follows (`m` stayed 0, World.step() then failed for want of the saved record and the warm
    start was lost).  The step with the limits enforced equals the oracle's; switching them off again restores the first result."""
    import torch
    import nimblephysics_amd as na
    from oracle import OracleWorld
    from test_gpu_joint_limits import _states
    from util import limited_arm, rel_err
    md_on = limited_arm(enforce=True)
    s, a = _states(md_on, 256, 3, at_limit=0.8)
    world = na.World(limited_arm(enforce=False), device="cuda:0")
    assert world.m == 0 and not any(world.getPositionLimitEnforced().values())
    st = torch.tensor(s, device="cuda:0"); at = torch.tensor(a, device="cuda:0")
    world.setState(st); world.setAction(at); world.step()
    free = world.getState().cpu().numpy()
    assert rel_err(free, OracleWorld(limited_arm(enforce=False)).step_batch(s, a, threads=4)["next"]) < 1e-9
    world.setPositionLimitEnforced(True)
    assert world.m == 25 and all(world.getPositionLimitEnforced().values())
    world.setState(st); world.setAction(at); world.step()                      # (want_saved = False: the scratch record must exist now)
    assert (world.last_status.cpu().numpy() & 0x400).astype(bool).mean() > 0.5
    assert world.lcp_cache is not None and tuple(world.lcp_cache.shape) == (25, 256)   # the warm start of the next step
    ref = OracleWorld(md_on).step_batch(s, a, threads=4)
    assert rel_err(world.getState().cpu().numpy(), ref["next"]) < 1e-7
    assert np.abs(world.getState().cpu().numpy() - free).max() > 1e-3           # the limits do something
    world.setPositionLimitEnforced(False)
    assert world.m == 0
    world.setState(st); world.setAction(at); world.step()
    assert np.array_equal(world.getState().cpu().numpy(), free)


def test_a_refused_setter_leaves_the_world_as_it_was():
    """A finite limit on a coordinate of a free-joint root cannot become an LCP row (nbl_model_create refuses): the setter raises, the
    flags are rolled back and the World keeps working on its old handle (it was left without one)."""
    import torch
    import nimblephysics_amd as na
    I = (0.01, 0.01, 0.01, 0, 0, 0)
    bodies = [na.BodySpec("root", -1, "free", "root_joint", mass=1.0, inertia=I, pos_lo=(-1.0,) * 6, pos_hi=(1.0,) * 6),
              na.BodySpec("arm", 0, "revolute", "hinge", axis=(0, 0, 1), T_pj=na.make_transform((0.2, 0, 0)), mass=0.5, inertia=I, pos_lo=(-0.5,), pos_hi=(0.5,))]
    world = na.World(na.ModelDescription("floating", bodies), device="cuda:0")
    s = torch.tensor(np.random.default_rng(1).normal(0, 0.1, (8, 14)), device="cuda:0"); a = torch.zeros((8, 7), dtype=torch.float64, device="cuda:0")
    world.setState(s); world.setAction(a); world.step()
    before = world.getState().clone()
    with pytest.raises(na.NimbleAmdError):
        world.setPositionLimitEnforced(True)
    assert not any(world.getPositionLimitEnforced().values())
    world.setState(s); world.setAction(a); world.step()
    assert torch.equal(world.getState(), before)
    world.setPositionLimitEnforced(True, joints=["hinge"])                      # the hinge alone is fine
    assert world.getPositionLimitEnforced()["hinge"] and world.m == 25


def test_self_collision_check_toggled_on_a_live_world():
    """World.setSelfCollisionCheck: the folding arm of tests/util.py touches itself only once its skeleton checks self-collisions."""
    import torch
    import nimblephysics_amd as na
    from oracle import OracleWorld
    from util import folding_arm, rel_err
    rng = np.random.default_rng(5)
    B = 128
    q = np.stack([rng.normal(0, 0.2, B), 2.1 + rng.normal(0, 0.03, B), 1.9 + rng.normal(0, 0.03, B)], 1)
    s = np.concatenate([q, rng.normal(0, 0.3, (B, 3))], 1); a = rng.normal(0, 0.1, (B, 3))
    world = na.World(folding_arm(self_collision=False), device="cuda:0")
    st = torch.tensor(s, device="cuda:0"); at = torch.tensor(a, device="cuda:0")
    world.setState(st); world.setAction(at); world.step()
    assert not (world.last_status.cpu().numpy() & 0x1).any()
    world.setSelfCollisionCheck(True)
    world.setState(st); world.setAction(at); world.step()
    status = world.last_status.cpu().numpy().astype(np.uint32)
    ref = OracleWorld(folding_arm(self_collision=True)).step_batch(s, a, threads=4)
    assert (status & 0x1).mean() > 0.2 and np.array_equal(status & 0x1, ref["status"] & 0x1)
    assert rel_err(world.getState().cpu().numpy(), ref["next"]) < 1e-7
    world.setSelfCollisionCheck(False)
    world.setState(st); world.setAction(at); world.step()
    assert not (world.last_status.cpu().numpy() & 0x1).any()
