"""`timestep(world, state, action)`: the drop-in autograd surface.

Mirrors python/nimblephysics/timestep.py:13-69 of the reference: same function name, argument order
and return arity; `TimestepLayer.forward(ctx, world, state, action, mass)` / `backward(ctx, grad)`
returning `(None, dState, dAction, dMass|None)`.

What changed, and only this: `state` is `[B, 2n]` and `action` `[B, k]` (a 1-D tensor is one world,
exactly the reference's shapes), the work happens in hand-written HIP kernels for B worlds at once,
and the result lives on the device of the input (CPU inputs are moved over and the result moved
back, so a reference script keeps working unchanged).
"""
from typing import Optional

import torch

from .world import World


class TimestepLayer(torch.autograd.Function):
    """A single differentiable timestep of B worlds as a PyTorch layer."""

    @staticmethod
    def forward(ctx, world: World, state: torch.Tensor, action: torch.Tensor, mass: Optional[torch.Tensor]):
        ctx.use_mass = mass is not None
        if ctx.use_mass:
            # world.setMasses(mass): the mass vector is a property of the (shared) model, one value for the whole batch
            if mass.dim() != 1 or mass.shape[0] != world.getMassDims():
                raise ValueError(f"mass must be a 1-D tensor of world.getMassDims() = {world.getMassDims()} entries")
            world.setMasses(mass)
            ctx.mass_device = mass.device
        one_d = state.dim() == 1
        in_device = state.device
        s = world._prep(state, 2 * world.n, "setState")      # world.setState(state)
        a = world._prep(action, world.k, "setAction")        # world.setAction(action)
        s_soa, a_soa = world.to_soa(s), world.to_soa(a)
        nxt, saved, status = world.step_soa(s_soa, a_soa, want_saved=True)   # nimble.neural.forwardPass(world)
        ctx.world = world
        ctx.saved_record = saved                              # the BackpropSnapshot
        ctx.one_d = one_d
        ctx.in_device = in_device
        ctx.action_device = action.device
        world._state = nxt
        world._action = a_soa
        out = world.from_soa(nxt)                             # torch.tensor(world.getState())
        if one_d:
            out = out[0]
        if in_device.type == "cpu" and out.device.type == "cuda":
            return world._to_host(out)                        # (pinned, asynchronous, one synchronisation: World._to_host)
        return out.to(in_device)

    @staticmethod
    def backward(ctx, grad_state):
        world: World = ctx.world
        g = grad_state.detach()
        if ctx.one_d:
            g = g.unsqueeze(0)
        g = world._prep(g, 2 * world.n, "backprop").contiguous()
        gs, ga = world.backward_soa(ctx.saved_record, world.to_soa(g))     # snapshot.backpropState(world, grad)
        d_state, d_action = world.from_soa(gs), world.from_soa(ga)
        d_mass = None
        if ctx.use_mass:
            # grads.lossWrtMass; the mass vector is shared by the B worlds, so their gradients add up
            d_mass = world.backward_inertia_soa(ctx.saved_record, g.shape[0]).sum(dim=1).to(ctx.mass_device)
        if ctx.one_d:
            d_state, d_action = d_state[0], d_action[0]
        if ctx.in_device.type == "cpu" and ctx.action_device.type == "cpu" and d_state.device.type == "cuda":
            d_state, d_action = world._to_host(d_state, d_action)
            return None, d_state, d_action, d_mass
        return None, d_state.to(ctx.in_device), d_action.to(ctx.action_device), d_mass


def timestep(world: World, state: torch.Tensor, action: torch.Tensor, mass: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Forward pass on `world`, storing what the backward pass needs (reference: timestep.py:63-69).  A world whose reference layout has
    coordinates of immobile skeletons (world.ref_layout, ref_layout.py) takes and returns the REFERENCE's [q; v]: those coordinates
    pass through unchanged (identity rows of the vector-Jacobian product), forces on them do nothing (zero gradient)."""
    lay = world.ref_layout
    if lay is None:
        return TimestepLayer.apply(world, state, action, mass)
    k_ref = len(world._ref_action_map)
    if action.shape[-1] != k_ref:
        raise ValueError(f"World.setAction() called with a tensor of incorrect size {tuple(action.shape)}; expected [B, {k_ref}]")
    cols = torch.tensor(lay.action_columns(world._ref_action_map), dtype=torch.long, device=action.device)
    out_dev = TimestepLayer.apply(world, lay.restrict_state(state, "World.setState()"), action.index_select(-1, cols), mass)
    return lay.expand_state(out_dev, state)


class RolloutLayer(torch.autograd.Function):
    """T differentiable timesteps of B worlds in one call (the loop of dart/trajectory/SingleShot.cpp:539-598 over
    forwardPass, and of backpropGradientWrt over backprop, kept on the device)."""

    @staticmethod
    def forward(ctx, world: World, state0: torch.Tensor, actions: torch.Tensor, warm_start: bool, mass: Optional[torch.Tensor] = None,
                checkpoint_every: int = 0):
        ctx.use_mass = mass is not None
        if ctx.use_mass:
            if mass.dim() != 1 or mass.shape[0] != world.getMassDims():
                raise ValueError(f"mass must be a 1-D tensor of world.getMassDims() = {world.getMassDims()} entries")
            world.setMasses(mass)                                     # constant along the trajectory
            ctx.mass_device = mass.device
        in_device = state0.device
        s = world._prep(state0, 2 * world.n, "setState")
        if actions.dim() != 3 or actions.shape[0] != s.shape[0] or actions.shape[2] != world.k:
            raise ValueError(f"rollout() wants actions [B, T, {world.k}]; got {tuple(actions.shape)}")
        B, T, k = actions.shape
        a = actions.detach().to(device=world.device, dtype=torch.float64)
        a_soa = a.permute(1, 2, 0).contiguous()                       # [T][k][B]
        states, saved, status = world.rollout_soa(world.to_soa(s), a_soa, want_saved=True, warm_start=warm_start,
                                                  checkpoint_every=checkpoint_every)
        ctx.world, ctx.saved_record, ctx.T = world, saved, T
        ctx.in_device, ctx.action_device = in_device, actions.device
        world._state = states[T]
        world.rollout_status = status
        world.rollout_record = saved          # what stays resident for the backward pass (a RolloutRecord when checkpointed)
        return states.permute(2, 0, 1).contiguous().to(in_device)     # [B, T+1, 2n]

    @staticmethod
    def backward(ctx, grad_states):
        world: World = ctx.world
        g = grad_states.detach().to(device=world.device, dtype=torch.float64).permute(1, 2, 0).contiguous()   # [T+1][2n][B]
        d_mass = None
        if ctx.use_mass:
            g0, ga, gm = world.rollout_backward_soa(ctx.saved_record, g, want_mass=True)
            d_mass = gm.sum(dim=1).to(ctx.mass_device)                # one mass vector for all worlds and steps
        else:
            g0, ga = world.rollout_backward_soa(ctx.saved_record, g)
        return None, world.from_soa(g0).to(ctx.in_device), ga.permute(2, 0, 1).contiguous().to(ctx.action_device), None, d_mass, None


def rollout(world: World, state0: torch.Tensor, actions: torch.Tensor, warm_start: bool = True,
            mass: Optional[torch.Tensor] = None, checkpoint_every: int = 0) -> torch.Tensor:
    """states[:, 0] = state0, states[:, t+1] = timestep(world, states[:, t], actions[:, t][, mass]).
    state0 [B, 2n], actions [B, T, k] -> states [B, T+1, 2n]; differentiable wrt state0, actions and (when given) the
    world's registered mass vector.  checkpoint_every = K > 0 keeps the backward records of K steps instead of T (a record is
    26.7 kB per world-step on Atlas-20 with contacts: world.saved_bytes(B) / B): the backward pass re-runs the other segments from their stored start states;
    the forward kernels are bit-reproducible, so the gradients are bit for bit those of checkpoint_every = 0."""
    lay = world.ref_layout
    if lay is None:
        return RolloutLayer.apply(world, state0, actions, warm_start, mass, checkpoint_every)
    # the reference's layout (ref_layout.py): the coordinates of immobile skeletons stay at state0's along the trajectory
    cols = torch.tensor(lay.action_columns(world._ref_action_map), dtype=torch.long, device=actions.device)
    dev = RolloutLayer.apply(world, lay.restrict_state(state0, "World.setState()"), actions.index_select(-1, cols), warm_start, mass, checkpoint_every)
    like = state0.unsqueeze(-2).expand(*dev.shape[:-1], state0.shape[-1])
    return lay.expand_state(dev, like)
