# stream slices of the default bench: 2 .. 6
for k in 2 3 4 5 6; do python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --streams $k 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $k M/s',round(j['value']/1e6,3),'ms',round(j['ms_per_step'],4))"; done
