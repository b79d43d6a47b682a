"""The multi-GPU layout end to end on what one GPU box offers (the reference's own parallelism is one cloned World per
shard, dart/trajectory/MultiShot.cpp:66-70, 183-200): TWO ranks under torch.distributed.run, each owning a contiguous shard of one batch of
the metric workload, a T-step differentiable rollout with ONE shared control sequence, and the ONE collective of the layout
(nimblephysics_amd.parallel.shared_parameter_grad: an all-gather of the per-rank partial gradients, summed in rank order).  Both ranks
sit on the box's only GPU, where RCCL refuses to form a communicator (two ranks, one device), so the group here is gloo; the RCCL
communicator is the same call on the driver's 8-GPU run.  Asserted: world for world the sharded run equals the unsharded one BIT FOR BIT
(states, status words, per-world gradients), the shared-control gradient equals the rank-ordered sum of the same shards of the unsharded
run bit for bit and is identical on both ranks, and exactly one collective is issued per backward pass."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_sharding_one_batch_equal_the_unsharded_run_bit_for_bit(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        import numpy as np, torch, torch.distributed as dist
        import nimblephysics_amd as na
        from nimblephysics_amd import parallel
        from util import contact_inputs
        dist.init_process_group("gloo")
        rank, ws = dist.get_rank(), dist.get_world_size()
        assert ws == 2
        calls = []
        real = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        dev = torch.device("cuda", 0)
        B, T = 512, 4
        md, s, a = contact_inputs("atlas20", B, 77, joint_noise=0.02, vel_noise=0.01, action_noise=0.1)   # the metric distribution

        def run(lo, hi):
            world = na.World(md, device=dev)
            st0 = world.to_soa(torch.tensor(s[lo:hi], device=dev)); act = world.to_soa(torch.tensor(a[:1].repeat(hi - lo, 0), device=dev))
            states, sv, status = world.rollout_soa(st0, act, T=T, want_saved=True, warm_start=True)
            gst = torch.zeros_like(states); gst[-1] = 2.0 * states[-1]
            g0, ga = world.rollout_backward_soa(sv, gst)           # ga [T][k][B_local]: per-world gradient of the shared control sequence
            return states, status, g0, ga

        lo, hi = parallel.shard_range(B, rank, ws)
        states, status, g0, ga = run(lo, hi)
        shared = parallel.shared_parameter_grad(ga.sum(0))          # THE collective: [k] on every rank
        assert len(calls) == 1, calls
        # the unsharded run (every rank computes it: the comparison needs no second collective)
        S, ST, G0, GA = run(0, B)
        assert torch.equal(states, S[:, :, lo:hi]) and torch.equal(status, ST[:, lo:hi])
        assert torch.equal(g0, G0[:, lo:hi]) and torch.equal(ga, GA[:, :, lo:hi])
        parts = [GA[:, :, l:h].sum(0).sum(dim=1) for (l, h) in (parallel.shard_range(B, r, ws) for r in range(ws))]
        assert torch.equal(shared, torch.stack(parts).sum(dim=0))
        assert (status[0].cpu().numpy() & 2 == 0).mean() > 0.2     # the fallback cascade ran on part of every shard
        both = [torch.zeros_like(shared.cpu()) for _ in range(ws)]
        dist.all_gather(both, shared.cpu())
        assert torch.equal(both[0], both[1])
        dist.destroy_process_group()
        print("ok", rank, float(shared.abs().max()))
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29571", str(script)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2
