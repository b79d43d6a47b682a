"""Multi-GPU layout for the batched timestep: one process per GPU, worlds sharded, ONE collective.

Worlds are independent (the reference's own parallelism is one cloned World per task,
dart/trajectory/MultiShot.cpp:66-70), so the batch dimension is partitioned contiguously across
ranks and nothing crosses GPUs during the steps.  The only exchange is the reduction of the
loss-gradient with respect to parameters SHARED by all worlds (policy weights, a shared control
sequence, ...): each rank reduces its own worlds locally, then a single all-gather over
RCCL/xGMI (backend "nccl" on ROCm) moves the per-rank partials — O(parameters) doubles, latency
bound — and every rank sums them in the same order, so all ranks hold bit-identical gradients.
Called once per trajectory backward, never per step.
"""
from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` worlds owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_sum(partial: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of the per-rank partial gradients via ONE all-gather (deterministic summation order)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return partial.clone()
    ws = dist.get_world_size(group)
    flat = partial.contiguous().view(-1)
    # (a gloo group - the CPU tests, two ranks sharing one GPU in tests/test_gpu_sharded_step.py: RCCL refuses two ranks on one device -
    #  moves host memory: the O(parameters) partial takes the trip through the host there)
    via_host = flat.is_cuda and dist.get_backend(group) == "gloo"
    send = flat.cpu() if via_host else flat
    buf = torch.empty(ws * send.numel(), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(buf, send, group=group)   # concatenated form: valid for RCCL and gloo
    total = buf.view(ws, *partial.shape).sum(dim=0)         # every rank sums the partials in rank order: bit-identical results
    return total.to(partial.device) if via_host else total


def shared_parameter_grad(per_world_grad_soa: torch.Tensor, group=None) -> torch.Tensor:
    """per_world_grad_soa: [k][B_local] gradient of every local world wrt a parameter shared by all
    worlds (DOF-major as the kernels produce it).  Returns the global [k] gradient on every rank."""
    return allgather_sum(per_world_grad_soa.sum(dim=1), group)
