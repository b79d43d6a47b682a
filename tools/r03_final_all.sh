# Everything that is measured on the final build of round 3, in one call on the GPU box: the GPU suite, the profile + bench variants
# (tools/r03_measure.sh), both passes of the randomised soaks (tools/r03_final_soak.sh) and the large mixed-feature soak.
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_suite.log 2>&1; tail -1 gpurun_out/r03_gpu_suite.log
bash tools/r03_measure.sh > gpurun_out/r03_measure.log 2>&1; tail -9 gpurun_out/r03_measure.log | cut -c1-200
bash tools/r03_final_soak.sh 0 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_final_soak.log
bash tools/r03_final_soak.sh 100000 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_second_soak.log
for v in balls big multi; do echo "mix:$v $(python tools/soak_stress.py mix 200000 1000 256 $v 2>&1 | grep -v amdgpu.ids | grep "MISMATCH\|mix\|Error" | tail -10)"; done > gpurun_out/r03_mix_soak.log 2>&1
grep -c "MISMATCH': 0" gpurun_out/r03_final_soak.log gpurun_out/r03_second_soak.log gpurun_out/r03_mix_soak.log
grep -L "MISMATCH" /dev/null; grep "MISMATCH seed\|Error" gpurun_out/r03_final_soak.log gpurun_out/r03_second_soak.log gpurun_out/r03_mix_soak.log | head
