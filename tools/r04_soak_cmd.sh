set -u
mkdir -p gpurun_out
(python tools/soak_reference_files.py _refdata/skel 64; python tools/soak_reference_files.py _refdata/robots 64) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_reference_model_files.log
tail -3 gpurun_out/r04_reference_model_files.log | cut -c1-300
bash tools/final_soak.sh 0 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_final_soak.log
grep -c "MISMATCH': 0" gpurun_out/r04_final_soak.log; grep "MISMATCH seed\|Error" gpurun_out/r04_final_soak.log | head
cat gpurun_out/r04_final_soak.log | cut -c1-400
