# round 6, first GPU call: the phase table of the cascade on the build as round 5 left it, the new / tightened tests, the baseline bench
mkdir -p gpurun_out
python tools/cascade_timing.py 0.02 > gpurun_out/r06_cascade_phases_base.log 2>&1
python -m pytest tests/test_gpu_general.py -x -q -s -k "narrow_phase or turned or towers or three" > gpurun_out/r06_general_tests.log 2>&1
tail -5 gpurun_out/r06_general_tests.log
python bench.py --no-cpu-baseline > gpurun_out/r06_bench_base.json 2> gpurun_out/r06_bench_base.err
tail -1 gpurun_out/r06_bench_base.json | cut -c1-400
rocm-smi --showclocks --showpower > gpurun_out/r06_smi_base.txt 2>&1
