"""The weak-scaling curve of bench.py on ONE node, self-explaining: `bench.py --gpus N` for N in --gpus (default 1 2 4 8,
capped at the devices visible), every N through the launcher path (torch.distributed.run, one process per GPU, RCCL over xGMI), plus the
plain single-process launch of N = 1 as the anchor.  Prints ONE table - per-GPU rate, efficiency against the N = 1 launcher run, the
min / max time per step over the ranks, the size of the RCCL communicator, the hardware-queue setting - and a JSON record, and ASSERTS
that the N = 1 launcher run is within --tolerance (3 %) of the plain launch: a launcher path that costs throughput by itself (round 5:
RCCL's idle streams took two of the four hardware queues, 4.27 instead of 6.6 M/s until GPU_MAX_HW_QUEUES=8) would otherwise be read
as a scaling loss.

usage (GPU box):  python tools/scale_curve.py [--gpus 1 2 4 8] [--steps 20] [--warmup 5] [--batch 4096] [--out gpurun_out/scale.json]
No efficiency is reported to the driver: it computes its own from the per-N `value`s; this is the builder's / maintainer's instrument."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCH_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def run_bench(extra, timeout=1800):
    env = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError(f"bench.py {' '.join(extra)} failed ({p.returncode}):\n{p.stderr[-3000:]}")
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        raise RuntimeError(f"bench.py {' '.join(extra)}: expected one JSON line, got {len(lines)}:\n{p.stdout[-2000:]}")
    return json.loads(lines[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--tolerance", type=float, default=0.03, help="allowed relative gap between the plain N = 1 launch and the launcher path at N = 1")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    visible = torch.cuda.device_count()
    assert visible >= 1, "no GPU visible"
    ns = [n for n in args.gpus if n <= visible]
    skipped = [n for n in args.gpus if n > visible]
    common = ["--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch), "--no-cpu-baseline", "--no-single-stream", "--easy-noise", "0"]

    plain = run_bench(["--gpus", "1"] + common)
    rows = []
    for n in ns:
        d = run_bench(["--gpus", str(n), "--spawn"] + common)
        c = d["config"]
        assert d["n_gpus"] == n and c["rccl_world_size"] == n, (n, d["n_gpus"], c["rccl_world_size"])
        rows.append({"n_gpus": n, "value": d["value"], "per_gpu": d["value"] / n, "ms_per_step": d["ms_per_step"],
                     "ranks_ms_per_step": c["ranks_ms_per_step"], "rccl_world_size": c["rccl_world_size"],
                     "gpu_max_hw_queues": c["gpu_max_hw_queues"], "devices_visible": c["devices_visible"], "scaling": d["scaling"]})
    base = rows[0]["value"] / rows[0]["n_gpus"]      # per-GPU rate of the smallest launcher run (N = 1 when it was asked for)
    for r in rows:
        r["efficiency_vs_first"] = r["per_gpu"] / base

    print(f"plain launch (no launcher, no process group), N = 1: {plain['value'] / 1e6:.3f} M worlds*steps/s, {plain['ms_per_step']:.4f} ms/step")
    print(f"{'N':>2} {'M worlds*steps/s':>17} {'per GPU':>9} {'efficiency':>10} {'ms/step (max rank)':>19} {'(min rank)':>10} {'RCCL ranks':>10} {'HW queues':>9}")
    for r in rows:
        rk = r["ranks_ms_per_step"] or {"max": r["ms_per_step"], "min": r["ms_per_step"]}
        print(f"{r['n_gpus']:>2} {r['value'] / 1e6:>17.3f} {r['per_gpu'] / 1e6:>9.3f} {r['efficiency_vs_first']:>10.3f} {rk['max']:>19.4f} {rk['min']:>10.4f} "
              f"{r['rccl_world_size']:>10} {str(r['gpu_max_hw_queues']):>9}")
    if skipped:
        print(f"not run (only {visible} device(s) visible): N = {skipped}")
    rec = {"plain_n1": {"value": plain["value"], "ms_per_step": plain["ms_per_step"]}, "launcher": rows, "skipped": skipped,
           "steps": args.steps, "warmup": args.warmup, "worlds_per_gpu": args.batch}
    if rows and rows[0]["n_gpus"] == 1:
        gap = rows[0]["value"] / plain["value"] - 1.0
        rec["launcher_vs_plain_n1"] = gap
        print(f"launcher path at N = 1 against the plain launch: {gap * 100:+.2f} % (allowed: +-{args.tolerance * 100:.0f} %)")
    print(json.dumps(rec))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    if "launcher_vs_plain_n1" in rec:
        assert abs(rec["launcher_vs_plain_n1"]) <= args.tolerance, \
            f"the launcher path costs {rec['launcher_vs_plain_n1'] * 100:+.1f} % at N = 1: fix that before reading the curve"
    return rec


if __name__ == "__main__":
    main()
