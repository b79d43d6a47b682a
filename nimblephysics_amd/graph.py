"""HIP-graph capture of a forward+backward step (or of a whole rollout) for callers that replay the same shapes many times.

The library only issues capturable work on the caller's stream (kernel launches, event fork / join of its internal streams; no
allocation, no host synchronisation), so `torch.cuda.CUDAGraph` records it as it is.  Replaying removes the ~12 inter-kernel
launch gaps of a step: 0.557 -> 0.518 ms per B = 4096 step on one stream (tools/graph_capture_experiment.py).  With the batch
sliced over several streams eager launches are faster than a multi-branch graph, so this helper captures on
ONE stream."""
from __future__ import annotations

from typing import Optional

import torch

from .world import World


class GraphedStep:
    """state [2n][B], action [k][B], grad_next [2n][B] are STATIC device tensors: write new values into them (copy_), call
    replay(), read next_state / grad_state / grad_action (also static)."""

    def __init__(self, world: World, B: int, warmup: int = 3, loss_grad=None):
        dev = world.device
        self.world = world
        self.state = torch.zeros((2 * world.n, B), dtype=torch.float64, device=dev)
        self.action = torch.zeros((world.k, B), dtype=torch.float64, device=dev)
        self.grad_next = torch.zeros((2 * world.n, B), dtype=torch.float64, device=dev)
        self._loss_grad = loss_grad          # optional: next_state -> dL/dnext_state, traced into the graph
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._warmup = warmup
        self.next_state = self.grad_state = self.grad_action = self.status = None

    def _step(self):
        w = self.world
        w.reset_lcp_cache()
        nxt, saved, status = w.step_soa(self.state, self.action)
        g = self._loss_grad(nxt) if self._loss_grad is not None else self.grad_next
        gs, ga = w.backward_soa(saved, g)
        self.next_state, self.grad_state, self.grad_action, self.status = nxt, gs, ga, status

    def capture(self):
        stream = torch.cuda.Stream(device=self.world.device)
        stream.wait_stream(torch.cuda.current_stream(self.world.device))
        with torch.cuda.stream(stream):
            for _ in range(self._warmup):
                self._step()
        stream.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, stream=stream):
            self._step()
        torch.cuda.current_stream(self.world.device).wait_stream(stream)
        return self

    def replay(self):
        if self._graph is None:
            self.capture()
        self._graph.replay()
        return self.next_state, self.grad_state, self.grad_action


class GraphedRollout:
    """A T-step rollout (nbl_rollout_forward + nbl_rollout_backward, the host loop of dart/trajectory/SingleShot.cpp:539-598 kept on the
    device) captured in ONE HIP graph: the ~2 x T x 9 launches per slice of a pass are issued by the graph instead of by the host thread,
    which is what bounds the eager rollout (Atlas-20 B = 4096 T = 64: 4.6 -> 7.0 M world-steps/s; Atlas-33 B = 8192: 5.2 -> 6.2;
    tools/rollout_graph_experiment.py).  Replays are bit-identical to the eager calls (tests/test_gpu_rollout.py).

    state0 [2n][B], actions [T][k][B] (or [k][B] with shared_action: one block applied at every step) and grad_states [T+1][2n][B] are
    STATIC device tensors: copy_ new values into them, call replay(), read states / grad_state0 / grad_actions / status (static as well).
    The T backward records (saved_bytes(B) each) stay resident in the graph's private memory pool for the lifetime of this object."""

    def __init__(self, world: World, B: int, T: int, shared_action: bool = False, warm_start: bool = True, loss_grad=None, warmup: int = 2):
        dev = world.device
        self.world, self.T, self.warm_start = world, T, warm_start
        self.state0 = torch.zeros((2 * world.n, B), dtype=torch.float64, device=dev)
        self.actions = torch.zeros(((world.k, B) if shared_action else (T, world.k, B)), dtype=torch.float64, device=dev)
        self.grad_states = torch.zeros((T + 1, 2 * world.n, B), dtype=torch.float64, device=dev)
        self._loss_grad = loss_grad          # optional: states [T+1][2n][B] -> dL/dstates, traced into the graph
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._warmup = warmup
        self.states = self.grad_state0 = self.grad_actions = self.status = None

    def _pass(self):
        w = self.world
        states, saved, status = w.rollout_soa(self.state0, self.actions, T=self.T, want_saved=True, warm_start=self.warm_start)
        g = self._loss_grad(states) if self._loss_grad is not None else self.grad_states
        g0, ga = w.rollout_backward_soa(saved, g)
        self.states, self.grad_state0, self.grad_actions, self.status = states, g0, ga, status

    def capture(self):
        stream = torch.cuda.Stream(device=self.world.device)
        stream.wait_stream(torch.cuda.current_stream(self.world.device))
        with torch.cuda.stream(stream):
            for _ in range(self._warmup):
                self._pass()
        stream.synchronize()
        self.states = self.grad_state0 = self.grad_actions = self.status = None      # (give the warm-up records back before the capture allocates its own)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, stream=stream):
            self._pass()
        torch.cuda.current_stream(self.world.device).wait_stream(stream)
        return self

    def replay(self):
        if self._graph is None:
            self.capture()
        self._graph.replay()
        return self.states, self.grad_state0, self.grad_actions
