TAG=${1:-r04}
# The soak part of tools/final_all.sh alone (the kernels were not touched after the measurements).
set -u
mkdir -p gpurun_out
bash tools/final_soak.sh 0 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_final_soak.log
bash tools/final_soak.sh 100000 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_second_soak.log
for v in balls big multi; do echo "mix:$v $(python tools/soak_stress.py mix 200000 1000 256 $v 2>&1 | grep -v amdgpu.ids | grep "MISMATCH\|mix\|Error" | tail -10)"; done > gpurun_out/${TAG}_mix_soak.log 2>&1
grep -c "MISMATCH': 0" gpurun_out/${TAG}_final_soak.log gpurun_out/${TAG}_second_soak.log gpurun_out/${TAG}_mix_soak.log
grep "MISMATCH seed\|Error" gpurun_out/${TAG}_final_soak.log gpurun_out/${TAG}_second_soak.log gpurun_out/${TAG}_mix_soak.log | head
