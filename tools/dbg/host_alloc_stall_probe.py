"""Probe: does a process that has HIP open stall for ~90 ms every few dozen large CPU-tensor allocations on this box (no nimble code involved)?"""
import sys, time
import torch
S = torch.randn(4096, 40, dtype=torch.float64)
dev = "cuda:0"


def run(name, f, n=60):
    its = []
    for i in range(n):
        t0 = time.perf_counter(); f(); its.append((time.perf_counter() - t0) * 1e3)
    its_s = sorted(its)
    print(f"{name}: median {its_s[n // 2]:.3f} ms, max {its_s[-1]:.2f} ms, iterations above 20 ms: {sum(x > 20 for x in its)} of {n}", flush=True)


def cpu_only():
    xs = [S.clone() for _ in range(8)]
    return xs


run("CPU churn before HIP is initialised (8 clones of 1.3 MB)", cpu_only)
torch.zeros(1, device=dev); torch.cuda.synchronize()
run("CPU churn after HIP is initialised", cpu_only)
pin = torch.empty(S.shape, dtype=torch.float64, pin_memory=True)
d = S.to(dev)


def gpu_copies():
    pin.copy_(S); x = pin.to(dev, non_blocking=True); pin.copy_(x, non_blocking=True); torch.cuda.synchronize()


run("pinned H2D + D2H only (no CPU allocation)", gpu_copies)


def both():
    xs = [S.clone() for _ in range(8)]; gpu_copies(); return xs


run("CPU churn + pinned copies", both)


def kernels_and_churn():
    xs = [S.clone() for _ in range(8)]; y = d * 2.0; torch.cuda.synchronize(); return xs


run("CPU churn + one device kernel + synchronise", kernels_and_churn)
