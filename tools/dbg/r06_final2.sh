mkdir -p gpurun_out
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver.json 2>/dev/null
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r06_gpu_suite.log 2>&1; tail -1 gpurun_out/r06_gpu_suite.log
timeout 900 bash tools/final_soak.sh 500000 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_final_soak.log; grep -c "MISMATCH.: 0" gpurun_out/r06_final_soak.log; grep -c . gpurun_out/r06_final_soak.log
NBL_SOAK_SLOTS=64 timeout 900 bash tools/final_soak.sh 600000 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_general_final_soak.log; grep -c "MISMATCH.: 0" gpurun_out/r06_general_final_soak.log
tail -1 gpurun_out/r06_bench_default.json | cut -c1-400
