# the default bench a few times: value, ms per step, kernel averages
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_avg_ms']
print('M/s',round(j['value']/1e6,3),'ms',round(j['ms_per_step'],4),' '.join('%s=%.0f'%(n.replace('k_','').replace('_coop','').replace('contact_','c_'),v*1e3) for n,v in k.items()))"; done
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --streams 1 --batch 1024 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_avg_ms']
print('slice1 ms',round(j['ms_per_step'],4),' '.join('%s=%.0f'%(n.replace('k_','').replace('_coop','').replace('contact_','c_'),v*1e3) for n,v in k.items()))"
