mkdir -p gpurun_out/prof
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQC_INST|INST_LEVEL|SQ_INSTS_VALU\b|SQ_BUSY_CY|SQ_INST_CYCLES|SQ_WAIT_INST" | sort -u | head -40 > $REPO/gpurun_out/prof/avail_icache.txt
cat $REPO/gpurun_out/prof/avail_icache.txt | cut -c1-200
ARGS="--steps 8 --warmup 2 --no-cpu-baseline --easy-noise 0 --min-seconds 0 --no-single-stream --streams 1 --batch 1024"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $REPO/gpurun_out/prof/ic -- python $REPO/bench.py $ARGS > $REPO/gpurun_out/prof/ic.log 2>&1
tail -2 $REPO/gpurun_out/prof/ic.log | cut -c1-300
python - <<PY
import csv, glob, collections
f = glob.glob("$REPO/gpurun_out/prof/ic/**/*counter_collection.csv", recursive=True)
print(f)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen=set()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key=(k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k]+=1
for k, v in agg.items():
    n = cnt[k]
    print(f"{k:62s} launches {n:4d} " + " ".join(f"{c}={x/n:.3g}" for c, x in sorted(v.items())))
PY
