TAG=${1:-r04}
# Everything that is measured on the final build of a round, in one call on the GPU box: the GPU suite, the profile + bench variants
# (tools/measure.sh), both passes of the randomised soaks (tools/final_soak.sh) and the large mixed-feature soak.
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_suite.log 2>&1; tail -1 gpurun_out/${TAG}_gpu_suite.log
bash tools/measure.sh ${TAG} > gpurun_out/${TAG}_measure.log 2>&1; tail -9 gpurun_out/${TAG}_measure.log | cut -c1-200
bash tools/final_soaks_only.sh ${TAG}
