TAG=${1:-r06}
# Everything that is measured on the final build of a round, in one call on the GPU box: the GPU suite, the general-build tests verbosely, the
# profile + bench variants (tools/measure.sh) and both passes of the randomised soaks (tools/final_soak.sh: as built, and with every model forced
# onto the general instantiation of the contact stage).
set -u
mkdir -p gpurun_out
(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i -E "sclk|power|temp" | head -8) > gpurun_out/${TAG}_box_state.log
bash tools/measure.sh ${TAG} > gpurun_out/${TAG}_measure.log 2>&1; tail -32 gpurun_out/${TAG}_measure.log | cut -c1-200
# cycles per phase: the cascade of the 24-row build (tools/cascade_timing.py, needs tools/dbg/libnimble_amd_timing.so) and the general build's solve kernel
[ -f tools/dbg/libnimble_amd_timing.so ] && timeout 300 python tools/cascade_timing.py 0.02 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_cascade_phases.log
[ -f tools/dbg/libnimble_amd_gentiming.so ] && timeout 300 python tools/gen_timing.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_general_phases.log
timeout 900 bash tools/scale_curve.sh > gpurun_out/${TAG}_scale_curve.log 2>&1; tail -4 gpurun_out/${TAG}_scale_curve.log | cut -c1-200
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gpu_suite.log 2>&1; tail -1 gpurun_out/${TAG}_gpu_suite.log
timeout 600 python -m pytest tests/test_gpu_general.py -q -s 2>&1 | grep -v amdgpu.ids | cut -c1-600 > gpurun_out/${TAG}_gpu_general_tests.log; tail -1 gpurun_out/${TAG}_gpu_general_tests.log
timeout 600 bash tools/final_soak.sh 500000 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_final_soak.log; grep -c "MISMATCH.: 0" gpurun_out/${TAG}_final_soak.log
NBL_SOAK_SLOTS=64 timeout 600 bash tools/final_soak.sh 600000 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_general_final_soak.log; grep -c "MISMATCH.: 0" gpurun_out/${TAG}_general_final_soak.log
# the bench lines once more, now that profiles/pmc_traffic.json and fp64_flops.json of THIS build exist (tools/aggregate_profile.py needs the
# merged-back counters: run `python tools/aggregate_profile.py <tag> atlas20_contact@0.02` and then tools/dbg/r06_final2.sh for lines with `traffic` and `fp64`)
