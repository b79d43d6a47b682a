mkdir -p gpurun_out
python tools/gen_timing.py 1024
python -m pytest tests/test_gpu_general.py -x -q > gpurun_out/r06_gen_tests.log 2>&1; tail -4 gpurun_out/r06_gen_tests.log
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --max-contacts 24 --steps 8 --warmup 2 > gpurun_out/r06_gen_bench.json 2>/dev/null
python - <<PY
import json
j = json.loads(open("gpurun_out/r06_gen_bench.json").read().strip().splitlines()[-1])
print("general", round(j["value"]), j["ms_per_step"], {k: round(v * 1e3, 1) for k, v in j["roofline"]["kernels_avg_ms"].items()})
PY
