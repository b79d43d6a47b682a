# worlds-per-workgroup cap of the tree kernels (NBL_TREE_WPB): headline and kernel averages, alternating runs on one box
one() { python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 $2 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_avg_ms']
print('$1','M/s',round(j['value']/1e6,3),'ms',round(j['ms_per_step'],4),' '.join('%s=%.0f'%(n.replace('k_','').replace('_coop','').replace('contact_','c_'),v*1e3) for n,v in k.items()))"; }
for rep in 1 2; do
  for w in 0 1 2; do
    if [ $w = 0 ]; then unset NBL_TREE_WPB; else export NBL_TREE_WPB=$w; fi
    one "wpb=$w"
  done
done
unset NBL_TREE_WPB; one "B32768 wpb=0" "--batch 32768"
export NBL_TREE_WPB=1; one "B32768 wpb=1" "--batch 32768"
