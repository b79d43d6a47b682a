b() { python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --steps 8 --warmup 2 --max-contacts 24 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']), round(j['ms_per_step'],3), {k: round(v*1e3,1) for k,v in j['roofline']['kernels_avg_ms'].items()})"; }
for i in 1 2 3; do
echo "pool 1152"; NBL_LIB_PATH=tools/dbg/libgen_qr0.so b
echo "pool 1536"; NBL_LIB_PATH=tools/dbg/libgen_qr1.so b
done
