"""The reference's snapshot interface for the hot path, batched: `nimble.neural.forwardPass(world, idempotent=False)` ->
`BackpropSnapshot` -> `backpropState(world, nextTimestepStateLossGrad)` -> `LossGradientHighLevelAPI`
(python/_nimblephysics/simulation_and_neural/NeuralGlobalMethods.cpp:49-53, BackpropSnapshot.cpp:59-140, NeuralUtils.cpp:58-67;
dart/neural/NeuralUtils.cpp forwardPass, BackpropSnapshot.cpp:61-179).  Same names and argument meaning; every vector of the reference
is a [B, .] tensor of B worlds here (a 1-D tensor is one world).  The snapshot is the saved record of nbl_step_forward."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ._lib import NimbleAmdError
from .world import World


class LossGradient:
    """dart::neural::LossGradient (NeuralUtils.hpp): the per-component form `BackpropSnapshot.backprop` takes and returns."""

    def __init__(self, lossWrtPosition=None, lossWrtVelocity=None, lossWrtTorque=None, lossWrtMass=None):
        self.lossWrtPosition, self.lossWrtVelocity = lossWrtPosition, lossWrtVelocity
        self.lossWrtTorque, self.lossWrtMass = lossWrtTorque, lossWrtMass


class LossGradientHighLevelAPI:
    """dart::neural::LossGradientHighLevelAPI: what `backpropState` returns."""

    def __init__(self, lossWrtState, lossWrtAction, lossWrtMass):
        self.lossWrtState, self.lossWrtAction, self.lossWrtMass = lossWrtState, lossWrtAction, lossWrtMass


class BackpropSnapshot:
    """What one forward pass keeps for its backward pass (dart/neural/BackpropSnapshot.hpp).  Built by forwardPass()."""

    def __init__(self, world: World, state_soa, action_soa, next_soa, saved, status, one_d: bool):
        self._world, self._state, self._action, self._next = world, state_soa, action_soa, next_soa
        self._saved, self._status, self._one_d = saved, status, one_d
        self._jac = None

    # ---- backward pass ----
    def _out(self, t):
        return t[0] if self._one_d else t

    def backpropState(self, world: World, nextTimestepStateLossGrad: torch.Tensor, perfLog=None,
                      exploreAlternateStrategies: bool = False) -> LossGradientHighLevelAPI:
        """BackpropSnapshot::backpropState (BackpropSnapshot.cpp:142-179): the loss gradient with respect to this step's state, action
        and (registered) masses, clipped at the joint limits like the reference (clipLossGradientsToBounds, :425-479)."""
        self._check(world)
        if exploreAlternateStrategies:
            raise NimbleAmdError("exploreAlternateStrategies is outside the hot-path scope")
        lay = world.ref_layout
        g_full = nextTimestepStateLossGrad
        if lay is not None:      # the reference's layout: the rows of the immobile coordinates are the identity's (ref_layout.py)
            nextTimestepStateLossGrad = lay.restrict_state(g_full.detach(), "backpropState", check_frozen=False)
        g = world._prep(nextTimestepStateLossGrad, 2 * world.n, "backpropState")
        gs, ga = world.backward_soa(self._saved, world.to_soa(g))
        if lay is not None:
            gm = world.backward_inertia_soa(self._saved, g.shape[0]).sum(dim=1) if world.getMassDims() > 0 else torch.zeros(0, dtype=torch.float64, device=world.device)
            gfull = g_full.detach().to(device=world.device, dtype=torch.float64)
            gfull = gfull if gfull.dim() == 2 else gfull.unsqueeze(0)
            ds = lay.expand_grad_state(world.from_soa(gs), gfull)
            cols = torch.tensor(lay.action_columns(world._ref_action_map), dtype=torch.long, device=world.device)
            da = torch.zeros((g.shape[0], len(world._ref_action_map)), dtype=torch.float64, device=world.device).index_copy(1, cols, world.from_soa(ga))
            return LossGradientHighLevelAPI(self._out(ds), self._out(da), gm)
        gm = None
        if world.getMassDims() > 0:
            gm = world.backward_inertia_soa(self._saved, g.shape[0]).sum(dim=1)     # one mass vector shared by the worlds
        else:
            gm = torch.zeros(0, dtype=torch.float64, device=world.device)
        return LossGradientHighLevelAPI(self._out(world.from_soa(gs)), self._out(world.from_soa(ga)), gm)

    def backprop(self, world: World, thisTimestepLoss: Optional[LossGradient], nextTimestepLoss: LossGradient, perfLog=None,
                 exploreAlternateStrategies: bool = False) -> LossGradient:
        """BackpropSnapshot::backprop (:181-423) in its component form: nextTimestepLoss carries lossWrtPosition / lossWrtVelocity of
        the NEXT state; the result (also stored into thisTimestepLoss when given) carries lossWrtPosition / Velocity / Torque / Mass."""
        n = world.getNumDofs()
        gq = world._prep(nextTimestepLoss.lossWrtPosition, n, "backprop")
        gv = world._prep(nextTimestepLoss.lossWrtVelocity, n, "backprop")
        hl = self.backpropState(world, torch.cat([gq, gv], dim=1), perfLog, exploreAlternateStrategies)
        st = hl.lossWrtState if hl.lossWrtState.dim() == 2 else hl.lossWrtState[None]
        if self._one_d:
            st = st[0]
        out = thisTimestepLoss if thisTimestepLoss is not None else LossGradient()
        out.lossWrtPosition, out.lossWrtVelocity = st[..., :n], st[..., n:]
        # lossWrtTorque: every DOF in the reference, zero where the action space does not reach; here the action space's DOFs
        out.lossWrtTorque, out.lossWrtMass = hl.lossWrtAction, hl.lossWrtMass
        return out

    # ---- Jacobians of the step (dense, 2n backward passes: diagnostics, like the reference's getters) ----
    def _jacobians(self, world):
        self._check(world)
        if self._jac is None:
            self._jac = world.step_jacobians_soa(self._saved, self._status.shape[0])
        return self._jac

    def getStateJacobian(self, world: World) -> torch.Tensor:
        """[B, 2n, 2n]: d next_state / d state = [[posPos, velPos], [posVel, velVel]] (BackpropSnapshot.cpp:2889-2917)."""
        J = self._jacobians(world)[0].permute(2, 0, 1).contiguous()
        return self._out(world.ref_layout.state_jacobian(J) if world.ref_layout is not None else J)

    def getActionJacobian(self, world: World) -> torch.Tensor:
        """[B, 2n, k]: d next_state / d action = [[0], [forceVel]] on the action space (BackpropSnapshot.cpp:2919-2940)."""
        J = self._jacobians(world)[1].permute(2, 0, 1).contiguous()
        return self._out(world._ref_action_jacobian(J) if world.ref_layout is not None else J)

    def _block(self, world, rows, cols):
        n = world.getNumDofs()
        J = self._jacobians(world)[0].permute(2, 0, 1)
        if world.ref_layout is not None:
            J = world.ref_layout.state_jacobian(J.contiguous())
        r0, c0 = (0 if rows == "pos" else n), (0 if cols == "pos" else n)
        return self._out(J[:, r0:r0 + n, c0:c0 + n].contiguous())

    def getPosPosJacobian(self, world, perfLog=None): return self._block(world, "pos", "pos")
    def getVelPosJacobian(self, world, perfLog=None): return self._block(world, "pos", "vel")      # d next position / d velocity
    def getPosVelJacobian(self, world, perfLog=None): return self._block(world, "vel", "pos")      # d next velocity / d position
    def getVelVelJacobian(self, world, perfLog=None): return self._block(world, "vel", "vel")

    def getControlForceVelJacobian(self, world, perfLog=None) -> torch.Tensor:
        """d next velocity / d control force on the DOFs of the action space, [B, n, k]."""
        return self._out(self._jacobians(world)[1].permute(2, 0, 1)[:, world.n:, :].contiguous())

    # ---- what was recorded ----
    def getPreStepPosition(self): return self._out(self._world.from_soa(self._state)[:, :self._world.n])
    def getPreStepVelocity(self): return self._out(self._world.from_soa(self._state)[:, self._world.n:])
    def getPreStepTorques(self): return self._out(self._world.from_soa(self._action))               # on the action space
    def getPostStepPosition(self): return self._out(self._world.from_soa(self._next)[:, :self._world.n])
    def getPostStepVelocity(self): return self._out(self._world.from_soa(self._next)[:, self._world.n:])
    def getPostStepTorques(self): return self.getPreStepTorques()
    def getStatus(self): return self._out(self._status)                                              # NBL_ST_* word per world

    def _check(self, world):
        """The record is read against the model constants of the world that is passed in: the world that took the snapshot, or one
        with the same model on the same device (a `clone()`; the reference's snapshots are likewise used with clones of their world,
        MultiShot.cpp:66-70)."""
        if world is self._world:
            return
        same = (isinstance(world, World) and world.device == self._world.device and _same_model(world.model, self._world.model))
        if not same:
            raise NimbleAmdError("this snapshot was taken on another world")


def _same_model(a, b) -> bool:
    if a is b:
        return True
    if (tuple(a.gravity), a.dt, a.max_contacts, a.contact_clipping_depth, a.fallback_cfm, bool(a.penetration_correction)) != \
            (tuple(b.gravity), b.dt, b.max_contacts, b.contact_clipping_depth, b.fallback_cfm, bool(b.penetration_correction)):
        return False
    fa, fb = a.flat(), b.flat()
    if fa.keys() != fb.keys():
        return False
    return all(np.array_equal(np.asarray(fa[k]), np.asarray(fb[k])) for k in fa)


def forwardPass(world: World, idempotent: bool = False) -> BackpropSnapshot:
    """nimble.neural.forwardPass (dart/neural/NeuralUtils.cpp): one timestep from the world's current state and action, recording
    what the backward pass needs.  idempotent = False (the default, like the reference) leaves the world in the next state; True restores
    the state it had (RestorableSnapshot)."""
    if getattr(world, "_state", None) is None or getattr(world, "_action", None) is None:
        raise NimbleAmdError("forwardPass(): call world.setState() and world.setAction() first")
    state, action = world._state, world._action
    if state.shape[1] != action.shape[1]:
        raise NimbleAmdError(f"forwardPass(): state holds {state.shape[1]} worlds, action {action.shape[1]}")
    cache = world.lcp_cache
    nxt, saved, status = world.step_soa(state, action, want_saved=True)
    one_d = bool(getattr(world, "_one_d", False))
    if idempotent:
        world.lcp_cache = cache                     # the solver's warm start belongs to the world's state
    else:
        world._state = nxt
    return BackpropSnapshot(world, state, action, nxt, saved, status, one_d)
