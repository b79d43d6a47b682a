"""The reference's snapshot interface on the device (nimblephysics_amd/neural.py): the flow of the reference's own TimestepLayer
(python/nimblephysics/timestep.py:30-60) - world.setState / setAction / [setMasses] -> nimble.neural.forwardPass(world) -> world.getState()
-> snapshot.backpropState(world, grad) -> lossWrtState / lossWrtAction / lossWrtMass - against the oracle, plus the snapshot's getters."""
import numpy as np
import pytest

from util import contact_inputs

pytestmark = pytest.mark.gpu


def test_forward_pass_and_backprop_state_like_the_references_timestep_layer():
    import torch
    import nimblephysics_amd as nimble
    from oracle import OracleWorld
    B = 128
    md, s, a = contact_inputs("atlas20", B, 71)
    g = np.random.default_rng(72).normal(0, 1, s.shape)
    world = nimble.World(md, device="cuda:0")
    world.tuneMass(1, nimble.WrtMassBodyNodeEntryType.INERTIA_MASS)
    # --- TimestepLayer.forward ---
    world.setState(torch.tensor(s)); world.setAction(torch.tensor(a)); world.setMasses(world.getMasses())
    snapshot = nimble.neural.forwardPass(world)
    nxt = world.getState()                                   # the world has moved on (idempotent = False)
    # --- TimestepLayer.backward ---
    grads = snapshot.backpropState(world, torch.tensor(g))
    ref = OracleWorld(md).step_batch(s, a, g, threads=8)
    sc = lambda x: max(np.abs(x).max(), 1e-30)
    assert np.abs(nxt.cpu().numpy() - ref["next"]).max() / sc(ref["next"]) < 1e-7
    assert np.abs(grads.lossWrtState.cpu().numpy() - ref["grad_state"]).max() / sc(ref["grad_state"]) < 1e-7
    assert np.abs(grads.lossWrtAction.cpu().numpy() - ref["grad_action"]).max() / sc(ref["grad_action"]) < 1e-7
    assert grads.lossWrtMass.shape == (world.getMassDims(),) and float(grads.lossWrtMass.abs().max()) > 0
    # what the snapshot recorded
    n = world.getNumDofs()
    assert torch.equal(snapshot.getPreStepPosition().cpu(), torch.tensor(s[:, :n])) and torch.equal(snapshot.getPreStepVelocity().cpu(), torch.tensor(s[:, n:]))
    assert torch.equal(snapshot.getPostStepPosition(), nxt[:, :n]) and torch.equal(snapshot.getPostStepVelocity(), nxt[:, n:])
    assert torch.equal(snapshot.getPreStepTorques().cpu(), torch.tensor(a))
    # the component form (BackpropSnapshot::backprop)
    lg = snapshot.backprop(world, None, nimble.neural.LossGradient(lossWrtPosition=torch.tensor(g[:, :n]), lossWrtVelocity=torch.tensor(g[:, n:])))
    assert torch.equal(lg.lossWrtPosition, grads.lossWrtState[:, :n]) and torch.equal(lg.lossWrtVelocity, grads.lossWrtState[:, n:])
    assert torch.equal(lg.lossWrtTorque, grads.lossWrtAction)


def test_idempotent_forward_pass_leaves_the_world_where_it_was_and_one_world_is_a_vector():
    import torch
    import nimblephysics_amd as nimble
    md, s, a = contact_inputs("atlas20", 1, 73)
    world = nimble.World(md, device="cuda:0")
    world.setState(torch.tensor(s[0])); world.setAction(torch.tensor(a[0]))          # 1-D like the reference
    before = world.getState().clone()
    snap = nimble.neural.forwardPass(world, idempotent=True)
    assert torch.equal(world.getState(), before)
    snap2 = nimble.neural.forwardPass(world)                                            # same step, now the world moves
    assert not torch.equal(world.getState(), before)
    assert torch.equal(snap.getPostStepPosition(), snap2.getPostStepPosition()) and snap.getPostStepPosition().dim() == 1
    g = torch.ones(2 * world.getNumDofs(), dtype=torch.float64)
    r1, r2 = snap.backpropState(world, g), snap2.backpropState(world, g)
    assert r1.lossWrtState.dim() == 1 and torch.equal(r1.lossWrtState, r2.lossWrtState) and torch.equal(r1.lossWrtAction, r2.lossWrtAction)
    with pytest.raises(nimble.neural.NimbleAmdError):
        nimble.neural.forwardPass(nimble.World(md, device="cuda:0"))                  # no state / action set


def test_snapshot_jacobians_are_the_blocks_of_the_oracles_dense_jacobians():
    import torch
    import nimblephysics_amd as nimble
    from oracle import OracleWorld
    B = 8
    md, s, a = contact_inputs("atlas20", B, 74)
    world = nimble.World(md, device="cuda:0")
    world.setState(torch.tensor(s)); world.setAction(torch.tensor(a))
    snap = nimble.neural.forwardPass(world, idempotent=True)
    Js, Ja = snap.getStateJacobian(world).cpu().numpy(), snap.getActionJacobian(world).cpu().numpy()
    n = world.getNumDofs()
    ow = OracleWorld(md)
    for b in range(B):
        ow.step(s[b], a[b])
        Rs, Ra = ow.getStateJacobian(), ow.getActionJacobian()
        assert np.abs(Js[b] - Rs).max() <= 1e-7 * np.abs(Rs).max() and np.abs(Ja[b] - Ra).max() <= 1e-7 * max(np.abs(Ra).max(), 1e-30)
    assert np.array_equal(snap.getPosPosJacobian(world).cpu().numpy(), Js[:, :n, :n]) and np.array_equal(snap.getVelPosJacobian(world).cpu().numpy(), Js[:, :n, n:])
    assert np.array_equal(snap.getPosVelJacobian(world).cpu().numpy(), Js[:, n:, :n]) and np.array_equal(snap.getVelVelJacobian(world).cpu().numpy(), Js[:, n:, n:])
    assert np.array_equal(snap.getControlForceVelJacobian(world).cpu().numpy(), Ja[:, n:, :])


def test_a_snapshot_serves_a_clone_of_its_world_and_refuses_another_model():
    """BackpropSnapshot._check: the world that took the snapshot or one with the same model (a clone(): the reference hands its
    snapshots to clones of the world, MultiShot.cpp:66-70); a world of another model raises NimbleAmdError, not AttributeError."""
    import torch
    import nimblephysics_amd as nimble
    md, s, a = contact_inputs("atlas20", 16, 75)
    world = nimble.World(md, device="cuda:0")
    world.setState(torch.tensor(s)); world.setAction(torch.tensor(a))
    snap = nimble.neural.forwardPass(world, idempotent=True)
    g = torch.tensor(np.random.default_rng(76).normal(0, 1, s.shape))
    r1 = snap.backpropState(world, g)
    r2 = snap.backpropState(world.clone(), g)
    assert torch.equal(r1.lossWrtState, r2.lossWrtState) and torch.equal(r1.lossWrtAction, r2.lossWrtAction)
    other = nimble.World(nimble.atlas("atlas20", ground=False), device="cuda:0")
    with pytest.raises(nimble.neural.NimbleAmdError):
        snap.backpropState(other, g)
    with pytest.raises(nimble.neural.NimbleAmdError):
        snap.getStateJacobian(other)
