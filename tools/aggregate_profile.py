"""Aggregate the rocprofv3 outputs of tools/profile.sh (merged back under gpurun_out/prof/) into profiles/:
   profiles/<tag>_kernel_stats_<workload>.csv   (copy of the --stats kernel summary)
   profiles/pmc_traffic.json                    (per-kernel HBM-side bytes per launch = (FETCH_SIZE + WRITE_SIZE) kB x 1024)
usage: python tools/aggregate_profile.py <tag> <workload>"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload = sys.argv[1], sys.argv[2]
P = os.path.join(ROOT, "gpurun_out", "prof")


def find(sub, pat):
    fs = glob.glob(os.path.join(P, f"{tag}_{sub}", "**", pat), recursive=True)
    if not fs:
        raise SystemExit(f"no {pat} under {tag}_{sub}")
    return fs[0]


stats = find("stats", "*kernel_stats.csv")
dst = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{workload}.csv")
shutil.copy(stats, dst)
print("copied", dst)


def counter(sub, name):
    f = find(sub, "*counter_collection.csv")
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("nbl::", "").replace("void ", "")
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


fetch, cf = counter("fetch", "FETCH_SIZE")
write, cw = counter("write", "WRITE_SIZE")
tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
out = json.load(open(tf)) if os.path.exists(tf) else {}
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (TCC slots), kB units x 1024, per launch, "
                "tools/profile.sh + tools/aggregate_profile.py. gfx950 FETCH_SIZE is calibrated (x2) only for 16-byte-wide coalesced "
                "streams (MI355X_MICROARCH.md, HBM); these kernels issue 8-byte loads, so the raw counter is reported uncorrected.")
out[workload] = {k: (fetch.get(k, 0) + write.get(k, 0)) * 1024 for k in sorted(set(fetch) | set(write)) if k.startswith("k_")}
out.setdefault("_detail", {})[workload] = {k: {"fetch_kb_per_launch": fetch.get(k, 0), "write_kb_per_launch": write.get(k, 0),
                                               "launches_profiled": cf.get(k, 0)} for k in out[workload]}
out["_tag"] = tag
json.dump(out, open(tf, "w"), indent=1)
print(json.dumps(out[workload], indent=1))
print("total per step launch (MB):", sum(v for k, v in out[workload].items() if k != "k_transpose") / 1e6)


# ---- VALU utilisation / occupancy passes (optional) ----
try:
    va, _ = counter("valu", "SQ_INSTS_VALU")
    wv, _ = counter("valu", "SQ_WAVES")
    ut, _ = counter("util", "VALUUtilization")
    vb, _ = counter("util", "VALUBusy")
    oc, _ = counter("util", "MeanOccupancyPerCU")
    vout = {k: {"valu_insts_per_launch": va.get(k), "waves": wv.get(k), "VALUUtilization_pct(lanes active)": round(ut.get(k, 0), 1),
                "VALUBusy_pct": round(vb.get(k, 0), 1), "MeanOccupancyPerCU": round(oc.get(k, 0), 2)} for k in sorted(va) if k.startswith("k_")}
    json.dump({"_note": "rocprofv3 --pmc passes of tools/profile.sh (SQ_INSTS_VALU SQ_WAVES | VALUUtilization VALUBusy MeanOccupancyPerCU), "
                        "per-launch averages. VALUBusy_pct: share of cycles a SIMD issues a VALU instruction; VALUUtilization_pct: share of "
                        "the 64 lanes active in the VALU instructions issued; MeanOccupancyPerCU: resident waves per CU (max 32).",
               workload: vout}, open(os.path.join(ROOT, "profiles", f"{tag}_valu_utilisation.json"), "w"), indent=1)
    print("wrote", f"profiles/{tag}_valu_utilisation.json")
except SystemExit as e:
    print("no VALU passes:", e)
