mkdir -p gpurun_out
b() { python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --steps 8 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']), round(j['ms_per_step'],3), {k: round(v*1e3,1) for k,v in j['roofline']['kernels_avg_ms'].items()})"; }
python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -2
echo c16 on general; NBL_MIN_VARIANT=2 b --max-contacts 16
echo c24 general; b --max-contacts 24
echo c8 on general; NBL_MIN_VARIANT=2 b
echo c32 general; b --max-contacts 32
