#!/bin/bash
# After `bash tools/final_all.sh <tag>` on the GPU box (gpurun merges gpurun_out/ back): aggregate the counters into profiles/ (hash-tied to the
# sources), copy the bench lines and logs, regenerate the kernel-resource tables.  Then run tools/dbg/r06_final2.sh-style bench lines once more
# on the GPU so that the default / driver lines carry `traffic` and `fp64` of THIS build.   usage: bash tools/refresh_profiles.sh <tag>
TAG=${1:-r06}
cd "$(dirname "$0")/.."
python tools/aggregate_profile.py ${TAG} atlas20_contact@0.02 | tail -2
for f in gpurun_out/${TAG}_bench_*.json; do tail -1 $f > profiles/$(basename $f); done
cp profiles/${TAG}_bench_default.json profiles/${TAG}_bench_line.json
for f in cascade_phases.log general_phases.log scale_curve.log gpu_suite.log gpu_general_tests.log box_state.log final_soak.log general_final_soak.log; do cp gpurun_out/${TAG}_$f profiles/${TAG}_$f; done
bash tools/kernel_resources.sh > profiles/${TAG}_kernel_resources.txt 2>&1 &
bash tools/kernel_resources.sh -DNBL_MAXC=16 > profiles/${TAG}_kernel_resources_48rows.txt 2>&1 &
bash tools/kernel_resources.sh -DNBL_MAXC=64 > profiles/${TAG}_kernel_resources_general.txt 2>&1 &
wait
python -c "import bench; print('csrc_sha16', bench.csrc_sha16())"; cat gpurun_out/prof/${TAG}_csrc_sha16.txt
