python bench.py --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err; tail -5 /tmp/b.err; python - <<'PY'
import json
try:
    j=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); print(j["value"], {k:(round(v["value"]/1e6,2) if isinstance(v,dict) and "value" in v else None) for k,v in j.get("secondary",{}).items()})
except Exception as e: print("no line", e)
PY
