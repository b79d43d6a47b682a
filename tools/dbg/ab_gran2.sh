one() { python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 $2 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1','M/s',round(j['value']/1e6,3),'ms',round(j['ms_per_step'],4))"; }
for wk in "--workload atlas20_freefall" "--workload atlas33_contact --batch 8192" "--workload atlas33_contact --rollout 64 --batch 8192" "--batch 8192" "--max-contacts 16"; do
  unset NBL_TREE_WPB NBL_DETECT_WL; one "default   $wk" "$wk"
  export NBL_TREE_WPB=1 NBL_DETECT_WL=8; one "wpb1 wl8  $wk" "$wk"
done
