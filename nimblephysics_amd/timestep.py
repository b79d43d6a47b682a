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
        if mass is not None:
            # reference: world.setMasses(mass); gradient wrt mass falls back to finite differences there
            # (Skeleton.cpp:1826-1829). Out of the hot-path scope (dart/neural row).
            raise NotImplementedError("timestep(..., mass=...) is outside the accelerated hot path")
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
        return out.to(in_device)

    @staticmethod
    def backward(ctx, grad_state):
        world: World = ctx.world
        g = grad_state.detach()
        if ctx.one_d:
            g = g.unsqueeze(0)
        g = g.to(device=world.device, dtype=torch.float64).contiguous()
        gs, ga = world.backward_soa(ctx.saved_record, world.to_soa(g))     # snapshot.backpropState(world, grad)
        d_state, d_action = world.from_soa(gs), world.from_soa(ga)
        if ctx.one_d:
            d_state, d_action = d_state[0], d_action[0]
        return None, d_state.to(ctx.in_device), d_action.to(ctx.action_device), None


def timestep(world: World, state: torch.Tensor, action: torch.Tensor, mass: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Forward pass on `world`, storing what the backward pass needs (reference: timestep.py:63-69)."""
    return TimestepLayer.apply(world, state, action, mass)
