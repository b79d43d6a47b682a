for k in 1 2 3 4; do python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --workload atlas33_contact --batch 8192 --streams $k 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_avg_ms']
print('streams $k','M/s',round(j['value']/1e6,3),'ms',round(j['ms_per_step'],4),' '.join('%s=%.0f'%(n.replace('k_','').replace('_coop','').replace('contact_','c_'),v*1e3) for n,v in k.items()))"; done
