# Everything that is measured on the final build of round 3, in one call on the GPU box: the GPU suite, the profile + bench variants
# (tools/r03_measure.sh), both passes of the randomised soaks (tools/r03_final_soak.sh) and the large mixed-feature soak.
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_suite.log 2>&1; tail -1 gpurun_out/r03_gpu_suite.log
bash tools/r03_measure.sh > gpurun_out/r03_measure.log 2>&1; tail -9 gpurun_out/r03_measure.log | cut -c1-200
bash tools/r03_final_soaks_only.sh
