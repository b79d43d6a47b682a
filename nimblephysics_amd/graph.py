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
