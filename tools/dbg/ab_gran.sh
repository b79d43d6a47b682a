# granularity of the forward tree kernel's workgroups: wavefronts per tree workgroup x worlds per narrow-phase workgroup
one() { python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 $2 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_avg_ms']
print('$1','M/s',round(j['value']/1e6,3),'ms',round(j['ms_per_step'],4),' '.join('%s=%.0f'%(n.replace('k_','').replace('_coop','').replace('contact_','c_'),v*1e3) for n,v in k.items()))"; }
for rep in 1 2; do
  unset NBL_TREE_WPB NBL_DETECT_WL; one "default      "
  export NBL_TREE_WPB=1 NBL_DETECT_WL=8; one "wpb=1 wl=8   "
  export NBL_TREE_WPB=1 NBL_DETECT_WL=16; one "wpb=1 wl=16  "
  export NBL_TREE_WPB=2 NBL_DETECT_WL=16; one "wpb=2 wl=16  "
done
unset NBL_TREE_WPB NBL_DETECT_WL; one "B32768 default" "--batch 32768"
export NBL_TREE_WPB=1 NBL_DETECT_WL=8; one "B32768 wpb=1 wl=8" "--batch 32768"
