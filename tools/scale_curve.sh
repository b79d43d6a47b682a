#!/bin/bash
# bench.py --gpus {1,2,4,8} -> one table (per-GPU rate, efficiency against N = 1, per-rank min / max, RCCL ranks) + the assertion that
# the launcher path at N = 1 is within 3 % of the plain launch.  See tools/scale_curve.py.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec python tools/scale_curve.py --out gpurun_out/scale_curve.json "$@"
