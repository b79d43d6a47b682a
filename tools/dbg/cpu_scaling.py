import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, oracle, tempfile
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")
so = os.path.join(tempfile.gettempdir(), "liboracle_native.so")
oracle.build(force=True, native=True, out=so)
for wl, noise in (("atlas20_freefall", 0.0), ("atlas20_contact", 0.002), ("atlas20_contact", 0.02)):
    md, s, a, _ = bench.make_workload(wl, 4096, 1000, noise)
    ow = oracle.OracleWorld(md, lib_path=so)
    g = 2 * s
    res = {}
    for th in (1, 16, 64, 128):
        n = 256 * th if th > 1 else 128
        S = np.tile(s, (n // len(s) + 1, 1))[:n]; A = np.tile(a, (n // len(a) + 1, 1))[:n]
        ow.step_batch(S[:th * 4], A[:th * 4], 2 * S[:th * 4], threads=th)
        t0 = time.perf_counter(); ow.step_batch(S, A, 2 * S, threads=th); dt = time.perf_counter() - t0
        res[th] = n / dt
    print(wl, noise, {k: round(v) for k, v in res.items()}, "scaling", {k: round(v / res[1], 1) for k, v in res.items()})
