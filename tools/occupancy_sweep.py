"""A/B of register caps (waves per SIMD the allocator must leave room for) per kernel: builds one library per variant
(here, hipcc cross-compiles) and, on the GPU box, runs the bench line of each.

  python tools/occupancy_sweep.py build            # -> build/variants/libnimble_amd_<tag>.so
  python tools/occupancy_sweep.py run [out.json]   # on the GPU: bench.py --no-cpu-baseline per variant
"""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "build", "variants")
VARIANTS = {
    "base": [],
    "stages2": ["-DNBL_W_STAGES=2"],
    "stages3": ["-DNBL_W_STAGES=3"],
    "stages4": ["-DNBL_W_STAGES=4"],
    "solve1": ["-DNBL_W_SOLVE=1"],
    "cfinal1": ["-DNBL_W_CFINAL=1"],
    "bwdb3": ["-DNBL_W_BWDB=3"],
    "rows4": ["-DNBL_W_ROWS=4"],
}


def build():
    os.makedirs(VDIR, exist_ok=True)

    def one(item):
        tag, flags = item
        out = os.path.join(VDIR, f"libnimble_amd_{tag}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", *flags,
               os.path.join(ROOT, "nimblephysics_amd", "csrc", "nimble_amd.hip"), "-o", out]
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return tag
    with ThreadPoolExecutor(4) as ex:
        for tag in ex.map(one, VARIANTS.items()):
            print("built", tag, flush=True)


def run(out_path, extra):
    res = {}
    for tag in VARIANTS:
        lib = os.path.join(VDIR, f"libnimble_amd_{tag}.so")
        if not os.path.exists(lib):
            continue
        env = dict(os.environ, NBL_LIB_PATH=lib)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--easy-noise", "0", *extra], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            res[tag] = {"error": p.stderr[-400:]}
            continue
        j = json.loads(line[-1])
        res[tag] = {"value": j["value"], "ms_per_step": j["ms_per_step"], "kernels_avg_ms": j["roofline"].get("kernels_avg_ms")}
        print(tag, round(j["value"] / 1e6, 3), "M/s", flush=True)
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "occupancy_sweep.json"), sys.argv[3:])
