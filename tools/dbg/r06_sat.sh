for st in 1 4; do
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --batch 32768 --streams $st 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_avg_ms']; print('streams $st', round(j['value']), j['ms_per_step'], {n: round(v*1e3,1) for n,v in sorted(k.items(), key=lambda x:-x[1])})"
done
