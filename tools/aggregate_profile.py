"""Aggregate the rocprofv3 outputs of tools/profile.sh (merged back under gpurun_out/prof/) into profiles/:
   profiles/<tag>_kernel_stats_<workload>.csv   (copy of the --stats kernel summary)
   profiles/pmc_traffic.json                    (per-kernel HBM-side bytes per launch = (FETCH_SIZE + WRITE_SIZE) kB x 1024)
   profiles/fp64_flops.json                     (fp64 flop per world-step COUNTED by the SQ instruction counters, pass `fp64`)
   profiles/<tag>_wave_cycles.json              (issue / wait split of the wave cycles, pass `wait`)
usage: python tools/aggregate_profile.py <tag> <workload>[@noise] [worlds per launch = 1024]"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload = sys.argv[1], sys.argv[2]
WPL = int(sys.argv[3]) if len(sys.argv) > 3 else 1024   # worlds covered by one kernel launch of the profiled bench
P = os.path.join(ROOT, "gpurun_out", "prof")
# the kernels the counters were taken on: tools/profile.sh writes the hash of csrc/ it saw on the GPU box (bench.csrc_sha16)
_tagfile = os.path.join(P, f"{tag}_csrc_sha16.txt")
CSRC_SHA16 = open(_tagfile).read().strip() if os.path.exists(_tagfile) else None


def find(sub, pat):
    fs = glob.glob(os.path.join(P, f"{tag}_{sub}", "**", pat), recursive=True)
    if not fs:
        raise SystemExit(f"no {pat} under {tag}_{sub}")
    return max(fs, key=os.path.getmtime)   # several runs of one tag leave one file each: the latest


stats = find("stats", "*kernel_stats.csv")
dst = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{workload.replace('@', '_noise')}.csv")
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
out.setdefault("_csrc_sha16", {})[workload] = CSRC_SHA16
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


# ---- fp64 flop count (pass `fp64`) ----
try:
    names = ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_MFMA_F64",
             "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"]
    C = {nm: counter("fp64", nm)[0] for nm in names}
    per = {}
    tot_flops = tot_wave_f64 = tot_mfma = 0.0
    lanes_w = 0.0
    for k in sorted(C["SQ_INSTS_VALU"]):
        if not k.startswith("k_") or k == "k_transpose":
            continue
        fma, add, mul, tr, mf = (C[nm].get(k, 0.0) for nm in names[:5])
        act = C["SQ_ACTIVE_INST_VALU"].get(k, 0.0)
        # SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = average active lanes per VALU instruction cycle (both in quad-cycles), max 64
        lanes = C["SQ_THREAD_CYCLES_VALU"].get(k, 0.0) / act if act else 0.0
        lanes = min(lanes, 64.0)
        flops = (2 * fma + add + mul + tr) * lanes + mf * 2048.0    # lane-level VALU flop of one launch (WPL worlds) + 2 x 16 x 16 x 4 per f64 MFMA
        per[k] = {"fma_f64": fma, "add_f64": add, "mul_f64": mul, "trans_f64": tr, "mfma_f64": mf, "valu_total": C["SQ_INSTS_VALU"].get(k, 0.0),
                  "avg_active_lanes": round(lanes, 1), "flops_per_world": flops / WPL}
        tot_flops += flops / WPL; tot_wave_f64 += (fma + add + mul + tr) / WPL; tot_mfma += mf / WPL
        lanes_w += lanes * (fma + add + mul + tr)
    ff = os.path.join(ROOT, "profiles", "fp64_flops.json")
    fo = json.load(open(ff)) if os.path.exists(ff) else {}
    fo["_note"] = ("rocprofv3 --pmc SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS,MFMA}_F64 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU (tools/profile.sh pass "
                   "`fp64`), per launch of WPL worlds.  flop = (2 FMA + ADD + MUL + TRANS wave-instructions) x average active lanes of the kernel's VALU "
                   "instructions (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU); summed over the kernels of one forward + one backward step.")
    fo[workload] = {"flops_per_world_step": tot_flops, "f64_wave_instr_per_world_step": tot_wave_f64, "mfma_f64_wave_instr_per_world_step": tot_mfma,
                    "valu_lane_utilisation": (lanes_w / (tot_wave_f64 * WPL) / 64.0) if tot_wave_f64 else None,
                    "counted_by": f"SQ_INSTS_VALU_*_F64 x active lanes, profiles tag {tag}", "csrc_sha16": CSRC_SHA16, "worlds_per_launch": WPL, "kernels": per}
    json.dump(fo, open(ff, "w"), indent=1)
    print("fp64 flop per world-step:", tot_flops, " f64 wave-instr per world-step:", tot_wave_f64, " MFMA f64:", tot_mfma)
except SystemExit as e:
    print("no fp64 pass:", e)

# ---- wave cycles: issue vs wait (pass `wait`) ----
try:
    names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_SALU", "SQ_INSTS_LDS"]
    C = {nm: counter("wait", nm)[0] for nm in names}
    wo = {}
    for k in sorted(C["SQ_WAVE_CYCLES"]):
        if not k.startswith("k_"):
            continue
        wc = C["SQ_WAVE_CYCLES"][k] or 1.0
        wo[k] = {nm: C[nm].get(k, 0.0) for nm in names}
        wo[k].update({"wait_any_frac": C["SQ_WAIT_ANY"].get(k, 0) / wc, "wait_inst_frac": C["SQ_WAIT_INST_ANY"].get(k, 0) / wc,
                      "active_frac": C["SQ_ACTIVE_INST_ANY"].get(k, 0) / wc})
    json.dump({"_note": "rocprofv3 --pmc pass `wait` of tools/profile.sh, per-launch averages; WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY "
                        "(issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles)", workload: wo},
              open(os.path.join(ROOT, "profiles", f"{tag}_wave_cycles.json"), "w"), indent=1)
    print("wrote", f"profiles/{tag}_wave_cycles.json")
except SystemExit as e:
    print("no wait pass:", e)
