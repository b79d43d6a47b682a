#!/bin/bash
# sweep worlds-per-workgroup for the tree kernels and the LDS-staged LCP kernels (bench.py, no CPU baseline)
mkdir -p gpurun_out
for tl in 64 16 8 4; do for ll in 16 8 4 2; do
  echo "== tree=$tl lcp=$ll" >> gpurun_out/sweep.log
  NBL_TREE_LANES=$tl NBL_LCP_LANES=$ll python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/sweep.log
done; done
cat gpurun_out/sweep.log
