for v in w2 w3 w2f; do
cp tools/dbg/libnimble_amd_$v.so nimblephysics_amd/libnimble_amd.so
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --max-contacts 24 --steps 8 --warmup 2 > /tmp/b.json 2>/dev/null
python - <<PY
import json
j = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("variant $v", round(j["value"]), j["ms_per_step"], round(j["roofline"]["kernels_avg_ms"]["k_contact_solve_coop"]*1e3))
PY
done
