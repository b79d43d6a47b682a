# round 6 iteration loop on the GPU box: the cascade's phase table, the bit-exactness self-tests + the parity tests that exercise the cascade, the quick bench
TAG=${1:-iter}
mkdir -p gpurun_out
python tools/cascade_timing.py 0.02 1024 > gpurun_out/${TAG}_cascade_phases.log 2>&1
python -m pytest tests/test_gpu_lcp_selftest.py tests/test_gpu_pinv_selftest.py tests/test_gpu_contact.py tests/test_gpu_contacts16.py -x -q > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
j = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print("${TAG}", round(j["value"]), j["ms_per_step"], {k: round(v * 1e3, 1) for k, v in j["roofline"]["kernels_avg_ms"].items()})
PY
grep -E "^  |Dantzig:|SLOW|stage [123] wave|PGS wavefront|slower of" gpurun_out/${TAG}_cascade_phases.log | cut -c1-200
