# A/B of the two-worlds-per-wavefront row kernel (NBL_ROWS_PACK=1: one world per wavefront), alternating runs
for B in 8192 32768; do for p in 2 1 2 1; do NBL_ROWS_PACK=$p python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --batch $B 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_avg_ms']
print('B',$B,'pack',$p,'M/s',round(j['value']/1e6,3),'ms',round(j['ms_per_step'],4),' '.join('%s=%.0f'%(n.replace('k_','').replace('_coop','').replace('contact_','c_'),v*1e3) for n,v in k.items()))"; done; done
