"""Multi-GPU layout on CPU: world_size-2 gloo processes exercise the sharding + the ONE collective
(all-gather of per-rank partial gradients, summed in rank order) used by bench.py --gpus N."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_batch():
    sys.path.insert(0, ROOT)
    from nimblephysics_amd.parallel import shard_range
    for total, ws in ((65536, 8), (10, 3), (7, 8), (4096, 1)):
        covered = []
        for r in range(ws):
            lo, hi = shard_range(total, r, ws)
            covered += list(range(lo, hi))
        assert covered == list(range(total))


def test_allgather_sum_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from nimblephysics_amd.parallel import shard_range, shared_parameter_grad
        dist.init_process_group("gloo")
        rank, ws = dist.get_rank(), dist.get_world_size()
        total, k = 10, 5
        g = torch.arange(total * k, dtype=torch.float64).reshape(total, k)     # per-world gradients of a shared parameter
        lo, hi = shard_range(total, rank, ws)
        local = g[lo:hi].t().contiguous()                                       # [k][B_local] as the kernels produce it
        out = shared_parameter_grad(local)
        assert torch.equal(out, g.sum(0)), (rank, out)
        # every rank holds bit-identical results
        buf = [torch.zeros_like(out) for _ in range(ws)]
        dist.all_gather(buf, out)
        assert all(torch.equal(b, buf[0]) for b in buf)
        dist.destroy_process_group()
        print("ok", rank)
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29561", str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2
