#!/bin/bash
# per-kernel times as a function of the batch size (latency-bound kernels stay flat, throughput-bound ones scale)
for bsz in 1024 2048 4096 8192 16384 32768; do
  python bench.py --batch $bsz --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_avg_ms']
print($bsz, round(d['value']), round(d['ms_per_step'],3), {a.replace('k_','').replace('contact_','c_'):round(b,3) for a,b in k.items()})"
done
