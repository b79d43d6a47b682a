#!/usr/bin/env python3
"""bench.py — worlds x timesteps / s, forward + backward, on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric config): 20-DOF Atlas (free root + 14 revolutes, arms welded) standing
on the ground box with 8 frictional foot-corner contacts (24 LCP rows), batch = 4096 worlds per GPU, pose
noise N(0, 0.02^2) as SURVEY.md 8(d) / BASELINE.md specify ("same distributions as cfg5"): about half of the
worlds leave LCP stage 0 and run the Dantzig / PGS cascade.  The same line carries, as `secondary.stage0_only`,
the rate on the easy distribution (noise 0.002: every world short-circuits at stage 0).
`--workload atlas33_contact --rollout 64 --batch 8192` is cfg5's per-GPU share (Atlas-33, T = 64 trajectory).
With `--gpus N` > 1 and no WORLD_SIZE in the environment the script re-launches itself under
torch.distributed.run (one process per GPU, RCCL).
A "step" is one differentiable timestep — forward (ABA, collision detection, LCP build by impulse tests,
stage-0 solve + standardisation) and backward (matrix-free adjoint incl. the contact terms) — of every
world of the batch through the C ABI (nbl_step_forward / nbl_step_backward), inputs resident in HBM in
the library's [dof][B] layout, cold LCP start every step (worst case: guess + solve).  The K timed steps
run back to back on the same synthetic batch; the gradient wrt the control vector shared by all worlds is
accumulated on device and reduced across GPUs with ONE all-gather (RCCL over xGMI) inside the timed
region.  Timing: barrier + synchronize on both sides, max over ranks.  Weak scaling: every GPU owns --batch
worlds.

Prints ONE JSON line (rank 0) with metric/value/roofline/cpu_baseline.  See DESIGN.md §Measurement.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# One process per GPU under torch.distributed.run: RCCL brings streams of its own.  They idle during the timed steps, but with the HIP runtime's
# default of FOUR hardware queues per process an idle stream still owns a queue and two of the four slice streams then share one: measured
# 4.27 against 6.66 M worlds*steps/s per GPU ("Throughput").  Eight queues give every stream its own again.  Read by the HIP
# runtime when it initialises, i.e. before the first torch.cuda call below; a caller's own setting wins.
if "WORLD_SIZE" in os.environ and not os.environ.get("GPU_MAX_HW_QUEUES"):
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # fp64 vector peak = 1/2 of the 157.3 TF fp32 vector peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)


def csrc_sha16():
    """Tag of the built kernels: sha256 over nimblephysics_amd/csrc/* and include/nimble_amd.h (sorted by name), first 16 hex digits.
    profiles/fp64_flops.json and pmc_traffic.json carry the tag of the sources they were counted on (tools/profile.sh writes it on the
    GPU box); counters of other sources are stale and are not reported."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "nimblephysics_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "nimble_amd.h"), "rb").read())
    return h.hexdigest()[:16]


def make_workload(name, B, seed, joint_noise):
    from util import cfg_inputs, contact_inputs
    if name == "atlas20_contact":
        md, s, a = contact_inputs("atlas20", B, seed, joint_noise=joint_noise, vel_noise=joint_noise / 2, action_noise=0.1)
        return md, s, a, ("Atlas 20-DOF (free root + 14 revolutes, arms welded) standing on the ground box, 8 frictional "
                          f"foot-corner contacts (24 LCP rows), pose q[0]=-pi/2 q[4]=-0.01 + N(0,{joint_noise}^2) joint noise")
    if name == "atlas33_contact":
        md, s, a = contact_inputs("atlas33", B, seed, joint_noise=joint_noise, vel_noise=joint_noise / 2, action_noise=0.1)
        return md, s, a, ("Atlas 33-DOF standing on the ground box, 8 frictional foot-corner contacts (24 LCP rows), cfg5 pose "
                          f"q[0]=-pi/2 q[4]=-0.01 + N(0,{joint_noise}^2) joint noise")
    key = {"atlas20_freefall": "atlas20", "atlas33_freefall": "atlas33", "cartpole": "cartpole"}[name]
    md, s, a = cfg_inputs(key, B, seed)
    return md, s, a, f"{key} without contact"


def _physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def _cpu_quota():
    """CPUs this container may actually use: the cgroup CPU quota (v2 cpu.max, v1 cfs_quota / cfs_period), None if unlimited.
    On the GPU boxes 256 logical CPUs are visible but the quota is 16: threads beyond it only time-share."""
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            return max(1, int(int(txt[0]) // int(txt[1])))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except Exception:
        pass
    return None


def cpu_baseline(md, state, action, target_seconds=20.0):
    """Time the CPU oracle (restated reference algorithm, scalar C++ -O3 -march=native) on this box's host cores on a bounded
    sample of the same workload: one cloned world per thread (the reference's own concurrency model, MultiShot.cpp:66-70),
    threads = PHYSICAL cores, at least 256 world-steps per thread (the batch is tiled to get there), per-thread bump arena
    instead of malloc in the timed path.  Reported, never used by the product."""
    import tempfile
    import oracle
    physical = _physical_cores()
    quota = _cpu_quota()
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = os.cpu_count() or physical
    threads = max(1, min(physical, affinity, quota or physical))       # the cores the box really gives this process
    logical = threads
    so = os.path.join(tempfile.gettempdir(), "liboracle_native.so")
    try:
        oracle.build(force=True, native=True, out=so)
        # keep the reference Dantzig next to the rebuilt oracle
        os.environ.setdefault("NBO_REF_DIR", os.path.join(os.path.dirname(oracle.__file__), "_ref"))
        ow = oracle.OracleWorld(md, lib_path=so)
    except Exception:
        ow = oracle.OracleWorld(md)
    g = 2.0 * state
    # one thread first: its rate sizes the sample
    n1 = min(128, len(state))
    ow.step_batch(state[:8], action[:8], g[:8], threads=1)          # warm-up
    t0 = time.perf_counter()
    ow.step_batch(state[:n1], action[:n1], g[:n1], threads=1)
    one = n1 / (time.perf_counter() - t0)

    def rate(nthreads, reps_n):
        per_thread = 256                                          # world-steps per thread and repetition
        n_sample = per_thread * nthreads
        reps_of_batch = (n_sample + len(state) - 1) // len(state)
        S = np.tile(state, (reps_of_batch, 1))[:n_sample]; A = np.tile(action, (reps_of_batch, 1))[:n_sample]; G = 2.0 * S
        ow.step_batch(S[:nthreads * 8], A[:nthreads * 8], G[:nthreads * 8], threads=nthreads)   # warm-up (thread creation, arenas)
        reps = []
        for _ in range(reps_n):
            t0 = time.perf_counter()
            ow.step_batch(S, A, G, threads=nthreads)
            reps.append(time.perf_counter() - t0)
        return n_sample / sorted(reps)[len(reps) // 2], n_sample

    val, n_sample = rate(threads, 3)
    extra = ""
    if logical != threads:
        val_l, _ = rate(logical, 1)
        extra = f"; {logical} logical threads: {val_l:.0f}/s"
    return {"value": val, "unit": "worlds*timesteps/s", "cores": threads, "kind": "port",
            "one_thread_value": one, "speedup_over_one_thread": val / one,
            "host": {"logical_cpus": os.cpu_count(), "physical_cores": physical, "cgroup_cpu_quota": quota},
            "sample": f"{n_sample} world-steps fwd+bwd ({n_sample // threads} per thread, the batch tiled), median of 3, {threads} threads = "
                      f"usable cores (min of physical cores {physical} and the container's CPU quota {quota}; one cloned world per thread, "
                      f"per-thread arena); 1 thread: {one:.1f}/s{extra}; restated reference "
                      "algorithm (oracle/, dense n x n Jacobians like BackpropSnapshot), not the upstream binary"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4096, help="worlds per GPU (weak scaling)")
    ap.add_argument("--workload", default="atlas20_contact")
    ap.add_argument("--joint-noise", type=float, default=0.02, help="pose noise of the metric distribution (SURVEY.md 8d: N(0, 0.02^2))")
    ap.add_argument("--easy-noise", type=float, default=0.002, help="pose noise of the secondary stage-0-only measurement (0 = skip it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=0, help="slices of the per-GPU batch, each a World on its own HIP stream (0 = auto: 4 from 4096 worlds, 2 from 2048)")
    ap.add_argument("--checkpoint-every", type=int, default=0, help="with --rollout: keep the backward records of K steps instead of T (the backward pass recomputes the other segments)")
    ap.add_argument("--rollout", type=int, default=0, help="diagnostic: one step = one pass of a T-step rollout fwd+bwd (nbl_rollout_*), value counts T*B worlds*steps per pass")
    ap.add_argument("--graph", action="store_true", help="with --rollout: capture the rollout's forward + backward pass in ONE HIP graph and time its replays")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostic: timed region without the per-kernel HIP events")
    ap.add_argument("--spawn", action="store_true", help="go through the self-launch path (torch.distributed.run, one process per GPU, RCCL) even for --gpus 1")
    ap.add_argument("--no-single-stream", action="store_true", help="skip the secondary one-launch-per-kernel measurement")
    ap.add_argument("--min-seconds", type=float, default=0.25, help="repeat the K-step timed region until this much time has been timed; the median repetition is reported")
    ap.add_argument("--max-reps", type=int, default=25)
    ap.add_argument("--max-contacts", type=int, default=0, help="diagnostic: contact slots per world of the model (0 = the workload's own, 8); 16 runs the "
                    "same worlds on the 48-row instantiation of the contact stage, 17..64 on the general one")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world_size and world_size > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")
    if (args.gpus > 1 or args.spawn) and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher (one process per GPU over RCCL) and relay rank 0's line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--spawn"]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    # one device per rank: LOCAL_RANK indexes the visible devices - unless the launcher masks them per rank (ROCR / HIP_VISIBLE_DEVICES: every
    # process then sees exactly one device, its own, as device 0)
    masked = world_size > 1 and torch.cuda.device_count() == 1
    dev_index = 0 if masked else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run (any N, also 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        # one process per GPU, one GPU per process: make a mis-launched job fail here, not report a wrong aggregate
        assert dist.get_world_size() == args.gpus, f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks"
        assert masked or torch.cuda.device_count() >= args.gpus, f"--gpus {args.gpus} but only {torch.cuda.device_count()} devices are visible"
        assert torch.cuda.current_device() == dev_index, f"rank {rank}: current device {torch.cuda.current_device()} != {dev_index} (LOCAL_RANK {local_rank})"
        assert dist.get_rank() == rank

    import nimblephysics_amd as na
    from nimblephysics_amd.parallel import shared_parameter_grad

    B = args.batch
    # The kernels of the step are latency / occupancy bound, so the forward of one slice of the batch overlaps the backward
    # of another when every slice owns a HIP stream (+16 % at B = 4096; the library cannot do this inside a call because a
    # call must join before it returns): one World per slice, all slices of one step issued before the next step.
    nstreams = args.streams if args.streams > 0 else (4 if B >= 4096 else (2 if B >= 2048 else 1))
    if args.rollout > 0:
        nstreams = 1
    per = (B + nstreams - 1) // nstreams
    bounds = [(i * per, min(B, (i + 1) * per)) for i in range(nstreams) if i * per < B]
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in bounds[1:]]

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def measure(noise, steps, warmup, kernel_timing, min_seconds=0.0, bounds=bounds, streams=streams, slices=None, one_handle=False):
        """W untimed + K timed fwd+bwd steps on a fresh synthetic batch of the workload at pose noise `noise`."""
        md, s_np, a_np, wl_desc = make_workload(args.workload, B, 1000 + rank, noise)
        if args.max_contacts > 0 and md.max_contacts:
            md.max_contacts = args.max_contacts
            build = "24-row" if args.max_contacts <= 8 else "48-row" if args.max_contacts <= 16 else "general (up to 192 rows, matrices in HBM scratch)"
            wl_desc += f" [max_contacts = {args.max_contacts}: the {build} build]"
        worlds = [na.World(md, device=dev) for _ in bounds]
        if slices is not None:
            for w_ in worlds:
                w_.set_slices(slices)
        world = worlds[0]
        k = world.k
        state0 = [w.to_soa(torch.tensor(s_np[lo:hi], device=dev)) for w, (lo, hi) in zip(worlds, bounds)]
        action = [w.to_soa(torch.tensor(a_np[lo:hi], device=dev)) for w, (lo, hi) in zip(worlds, bounds)]
        torch.cuda.synchronize(dev)

        graphed = None
        if args.rollout > 0 and args.graph:
            from nimblephysics_amd.graph import GraphedRollout
            graphed = GraphedRollout(world, B, args.rollout, shared_action=True, warm_start=True,
                                     loss_grad=lambda states: torch.cat([torch.zeros_like(states[:-1]), 2.0 * states[-1:]], 0))
            graphed.state0.copy_(state0[0]); graphed.actions.copy_(action[0])
            graphed.capture()

        deferred = None
        if one_handle:
            # ONE World handle in deferred-join mode (include/nimble_amd.h, ABI minor 5): the library runs every slice of a call on an
            # internal stream and does not join per call; the loss of a slice is enqueued on the slice's stream.  Caller-owned buffers,
            # reused every step: everything that touches the worlds of slice i runs on stream i, in order.
            world.set_deferred_join(True)
            sl = world.slices(B)
            n2_ = 2 * world.n
            deferred = dict(sl=sl, nxt=torch.empty((n2_, B), dtype=torch.float64, device=dev),
                            saved=torch.empty(world.saved_bytes(B), dtype=torch.uint8, device=dev),
                            status=torch.empty(B, dtype=torch.int32, device=dev),
                            cache=torch.empty((world.m, B), dtype=torch.float64, device=dev) if world.m > 0 else None,
                            g=torch.empty((n2_, B), dtype=torch.float64, device=dev), gs=torch.empty((n2_, B), dtype=torch.float64, device=dev),
                            ga=torch.empty((k, B), dtype=torch.float64, device=dev))
            torch.cuda.synchronize(dev)

        def run(T):
            if deferred is not None:
                d = deferred
                ga_sum = torch.zeros((k, B), dtype=torch.float64, device=dev)
                world.fork()                                   # (the slices' streams start behind what this stream holds: the zeroed sum)
                for _ in range(T):
                    world.step_into(state0[0], action[0], d["nxt"], d["saved"], d["status"], None, d["cache"])      # cold start: no warm start in
                    for stream, lo, hi in d["sl"]:
                        with torch.cuda.stream(stream):
                            torch.mul(d["nxt"][:, lo:hi], 2.0, out=d["g"][:, lo:hi])                               # d/ds' of |s'|^2
                    world.backward_into(d["saved"], d["g"], d["gs"], d["ga"])
                    for stream, lo, hi in d["sl"]:
                        with torch.cuda.stream(stream):
                            ga_sum[:, lo:hi] += d["ga"][:, lo:hi]                                                   # the control vector is shared by all steps
                world.join()
                return shared_parameter_grad(ga_sum), d["status"]                                                 # ONE all-gather per timed region
            ga_total = [torch.zeros((k, hi - lo), dtype=torch.float64, device=dev) for (lo, hi) in bounds]
            status = [None] * len(bounds)
            if args.rollout > 0 and args.graph:      # the same pass replayed from ONE captured HIP graph (nimblephysics_amd.graph.GraphedRollout)
                for _ in range(T):
                    graphed.replay()
                    ga_total[0] += graphed.grad_actions.sum(0)
                return shared_parameter_grad(ga_total[0]), graphed.status[0]
            if args.rollout > 0:      # cfg5-style: T-step trajectory, loss = |q_T|^2 + |v_T|^2, one shared control vector
                for _ in range(T):
                    states, sv, st_all = world.rollout_soa(state0[0], action[0], T=args.rollout, want_saved=True, warm_start=True, checkpoint_every=args.checkpoint_every)
                    gst = torch.zeros_like(states)
                    gst[-1] = 2.0 * states[-1]
                    g0, ga = world.rollout_backward_soa(sv, gst)
                    ga_total[0] += ga.sum(0)
                    status[0] = st_all[0]        # the cold-start step (later steps are warm-started and resolve at stage 0)
                return shared_parameter_grad(ga_total[0]), status[0]
            main = streams[0]
            for st in streams[1:]:
                st.wait_stream(main)
            for _ in range(T):
                for i, (w, st) in enumerate(zip(worlds, streams)):
                    with torch.cuda.stream(st):
                        w.reset_lcp_cache()                                        # cold start: guess + solve every step
                        nxt, sv, status[i] = w.step_soa(state0[i], action[i], want_saved=True)
                        gs, ga = w.backward_soa(sv, 2.0 * nxt)                     # d/ds' of |s'|^2
                        ga_total[i] += ga                                          # the control vector is shared by all steps
            for st in streams[1:]:
                main.wait_stream(st)
            return shared_parameter_grad(torch.cat(ga_total, 1)), torch.cat(status)   # ONE all-gather per timed region

        if warmup > 0:
            run(warmup)
        sync()
        # per-kernel HIP events on a sample of the timed steps (every 8th): events around all ~12 launches of every step cost 8 %
        timing_period = 8 if steps >= 16 else (4 if steps >= 8 else 1)
        world.set_timing(kernel_timing, timing_period)
        # The timed region = EXACTLY `steps` steps between barrier + synchronize on both sides, max over ranks.  A short region (the
        # driver's --steps 20 is 12 ms) is a noisy sample: it is repeated until `min_seconds` have been timed (every rank takes the
        # same decision from the max-over-ranks time) and the MEDIAN repetition is reported.
        reps, rank_min = [], []
        while True:
            t0 = time.perf_counter()
            grad, status = run(steps)
            sync()
            el = time.perf_counter() - t0
            if use_dist:
                t = torch.tensor([el], dtype=torch.float64, device=dev)
                tmin = t.clone()
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
                rank_min.append(float(tmin.item()))
                el = float(t.item())
            reps.append(el)
            assert torch.isfinite(grad).all()
            if sum(reps) >= min_seconds or len(reps) >= args.max_reps:
                break
        elapsed = sorted(reps)[len(reps) // 2]
        tm = world.get_timing()
        world.set_timing(False)
        st = status.cpu().numpy().astype(np.uint32)
        # (the handles - and the HIP streams they own - die with this frame: a later measurement starts with fresh streams.  The runtime maps
        #  streams to its four hardware queues in creation order; with the streams of an earlier handle still alive the two slices of a
        #  joined call shared a queue: `single_call` 3.85 instead of 5.9 M/s)
        n_slices = len(bounds) * max(1, world.slices_for(bounds[0][1] - bounds[0][0]))
        dims = {"n": world.n, "m": world.m, "saved_bytes_per_world": int(world._L.nbl_saved_bytes(world._h, B) // B)}
        del deferred, graphed, worlds, world
        return {"elapsed": elapsed, "reps": reps, "rank_min": rank_min, "status": st, "timing": tm, "timing_period": timing_period, "md": md, "s": s_np, "a": a_np,
                "desc": wl_desc, "dims": dims, "slices": n_slices}

    has_contact = args.workload.endswith("_contact")
    # The timed region is ONE model handle in deferred-join mode (round 6): `--streams k` (k > 0) goes back to round 5's
    # harness pattern - k Worlds, each on its own stream - which stays as a secondary figure (`four_handles`).
    one = args.streams == 0 and args.rollout == 0
    if one:
        R = measure(args.joint_noise, args.steps, args.warmup, not args.no_kernel_timing, args.min_seconds, [(0, B)], streams[:1], one_handle=True)
    else:
        R = measure(args.joint_noise, args.steps, args.warmup, not args.no_kernel_timing, args.min_seconds)
    easy = None
    if has_contact and args.easy_noise > 0 and args.easy_noise != args.joint_noise:
        # the easy distribution: every world resolves at LCP stage 0 (round 1's headline), half the steps, no kernel events
        easy = measure(args.easy_noise, max(1, args.steps // 2), min(args.warmup, 4), False)
    single = single_call = one_handle_deferred = None
    if has_contact and len(bounds) > 1 and not args.no_single_stream:
        # the same batch as ONE launch per kernel per step (no stream slices): what the chain costs without the overlap of the slices
        single = measure(args.joint_noise, max(1, args.steps // 2), min(args.warmup, 4), False, 0.0, [(0, B)], streams[:1], slices=1)
        # ... and as ONE call per step with the library's own slicing (what a caller gets who hands over the whole batch at once)
        single_call = measure(args.joint_noise, max(1, args.steps // 2), min(args.warmup, 4), False, 0.0, [(0, B)], streams[:1])
        # ... and round 5's harness pattern: one World per slice, each on its own HIP stream (what ONE handle in deferred-join mode replaces)
        if one:
            one_handle_deferred = measure(args.joint_noise, args.steps, min(args.warmup, 4), False, 0.0)
    host_tensors = None
    if has_contact and not args.no_single_stream and world_size == 1 and args.rollout == 0:
        # the reference's OWN calling convention (python/nimblephysics/timestep.py:31-40): float64 CPU tensors [B, 2n] / [B, k] in, CPU tensors
        # out, through the drop-in surface timestep() + .backward() of ONE World - pinned staging, asynchronous copies on the step's stream,
        # the two layout transposes, everything.  PCIe-inclusive: a secondary figure, never `value`.
        from nimblephysics_amd.timestep import timestep
        hw = na.World(R["md"], device=dev)
        hs = torch.tensor(R["s"], requires_grad=True); ha = torch.tensor(R["a"], requires_grad=True)
        hsteps = max(8, args.steps)
        # torch's CPU ops of this leg (copies of 1.3 MB, the loss) run on ONE intra-op thread: torch's default (one OpenMP thread per logical
        # CPU, 256 here, spinning after every op) overruns the container's CFS quota (16 CPUs on the GPU boxes of this pool) and the whole
        # process is throttled for the rest of the 100 ms period once every few steps (tools/dbg/host_alloc_stall_probe.py: cpu.stat
        # nr_throttled goes up, 0.39 M/s instead of 3.3); a caller in such a container sets OMP_NUM_THREADS to its quota for the same reason
        cpu_threads = torch.get_num_threads()
        torch.set_num_threads(1)
        for it in range(3 + hsteps):
            if it == 3:
                torch.cuda.synchronize(dev); t0h = time.perf_counter()
            hs.grad = None; ha.grad = None
            hw.reset_lcp_cache()
            hout = timestep(hw, hs, ha)
            (hout * hout).sum().backward()
        torch.cuda.synchronize(dev)
        host_tensors = {"elapsed": time.perf_counter() - t0h, "steps": hsteps}
        torch.set_num_threads(cpu_threads)
        del hw
    elapsed, st, tm, timing_period, md, s_np, a_np, wl_desc, dims = (R[x] for x in ("elapsed", "status", "timing", "timing_period", "md", "s", "a", "desc", "dims"))
    n = dims["n"]

    if rank == 0:
        units_per_step = B * world_size * max(1, args.rollout)
        total_units = units_per_step * args.steps
        value = total_units / elapsed
        m_rows = 24 if dims["m"] > 0 else 0
        kern = {kname: v["ms_sum"] / v["count"] for kname, v in tm["kernels"].items()}
        if not kern:   # no per-kernel events (--no-kernel-timing, or the rollout entry points, which own their streams): whole-step figures
            kern = {"whole_step": elapsed / args.steps * 1e3 / max(1, args.rollout)}
        # algorithmic HBM bytes per world-step fwd+bwd (fp64) = 104 n + 16 m.
        # Per launch of the step (all kernels of one forward + one backward): that figure x B worlds.
        dom = max(kern, key=kern.get)
        alg_step_bytes = (104 * n + 16 * m_rows) * B
        slices = R["slices"]   # the batch is processed as slices on overlapping HIP streams:
        alg_launch_bytes = alg_step_bytes / slices  # one kernel launch covers B / slices worlds (timed on slice 0)
        step_kernel_ms = sum(kern.values())
        # the dominant kernel is credited with the whole step's algorithmic traffic share it is responsible for:
        # state/cotangent I/O is spread over the kernels, so the conservative per-kernel figure is
        # (algorithmic bytes of one step) / (duration of the dominant kernel) -- an upper bound on its fraction.
        achieved = alg_launch_bytes / (kern[dom] * 1e-3) / 1e9
        traffic = None
        prof_key = f"{args.workload}@{args.joint_noise:g}"
        built = csrc_sha16()
        stale = []
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("_csrc_sha16", {}).get(prof_key) == built:
                    per_kernel = tj.get(prof_key, {})
                    traffic = next((v for kname, v in per_kernel.items() if kname.split("<")[0] == dom), None)   # template suffixes: k<false>
                elif prof_key in tj:
                    stale.append("pmc_traffic.json")
            except Exception:
                traffic = None
        # fp64 roofline: flop COUNTED by the SQ instruction counters (tools/profile.sh pass `fp64`, aggregated by
        # tools/aggregate_profile.py into profiles/fp64_flops.json): 2 x FMA + ADD + MUL + TRANS wave-instructions x active lanes.
        # Only counters taken on the kernels that are built right now are used (tag = hash of csrc/).
        fp64 = None
        ffile = os.path.join(ROOT, "profiles", "fp64_flops.json")
        if os.path.exists(ffile):
            try:
                fj = json.load(open(ffile)).get(prof_key)
                if fj and fj.get("csrc_sha16") == built:
                    fl = float(fj["flops_per_world_step"])
                    per_gpu_rate = value / world_size
                    fp64 = {"flops_per_world_step": fl, "counted_by": fj.get("counted_by"), "csrc_sha16": built,
                            "achieved_TFs": fl * per_gpu_rate / 1e12,
                            "peak_TFs": FP64_PEAK_TFLOPS, "frac": fl * per_gpu_rate / 1e12 / FP64_PEAK_TFLOPS,
                            "mfma_f64_wave_instr_per_world_step": fj.get("mfma_f64_wave_instr_per_world_step"),
                            "valu_lane_utilisation": fj.get("valu_lane_utilisation")}
                    # the dominant kernel on its own: its counted flop per launch / its average launch duration (HIP events)
                    kd = next((v for kname, v in fj.get("kernels", {}).items() if kname.split("<")[0] == dom), None)
                    if kd:
                        k_tfs = kd["flops_per_world"] * (B // slices) / (kern[dom] * 1e-3) / 1e12
                        fp64["dominant_kernel"] = {"kernel": dom, "flops_per_launch": kd["flops_per_world"] * (B // slices), "avg_launch_ms": kern[dom],
                                                   "achieved_TFs": k_tfs, "frac": k_tfs / FP64_PEAK_TFLOPS, "avg_active_lanes": kd.get("avg_active_lanes")}
                elif fj:
                    stale.append("fp64_flops.json")
            except Exception:
                fp64 = None
        hbm = {"kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
               "algorithmic_bytes_per_launch": alg_launch_bytes, "worlds_per_launch": B // slices,
               "algorithmic_bytes_per_step": alg_step_bytes, "avg_launch_ms": kern[dom],
               "whole_step_achieved_GBs": alg_step_bytes * max(1, args.rollout) / (elapsed / args.steps) / 1e9,
               "whole_step_frac": alg_step_bytes * max(1, args.rollout) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS}
        common = {"stream_slices": slices, "kernels_avg_ms": kern, "step_kernel_ms": step_kernel_ms, "timed_every_nth_step": timing_period,
                  "csrc_sha16": built, "stale_counter_files_ignored": stale}
        if fp64 is not None:
            # ~300 flop per algorithmic byte - the fp64 vector ALU is the roof that binds, so it is the primary fraction
            # (whole step: counted flop per world-step x measured rate); the HBM figures north_star asks for ride along under `hbm`.
            roof = {"bound": "fp64", "kernel": "whole step (all kernels of one forward + one backward)", "achieved": fp64["achieved_TFs"],
                    "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fp64["frac"], "traffic": traffic, "fp64": fp64, "hbm": hbm, **common,
                    "note": "fp64-ALU / latency bound (SURVEY.md 8d); peak = fp64 vector peak 78.6 TF (v_fma_f64 measured at 62.5 TF, "
                            "v_mfma_f64 at 49.0 TF on this chip: docs/profiles_history/r02c_fp64_rate.jsonl); flop counted by SQ_INSTS_VALU_*_F64"}
        else:
            roof = {"bound": "hbm", **hbm, "fp64": None, **common,
                    "note": "no fp64 flop count for the kernels built right now (profiles/fp64_flops.json is absent or was counted on other "
                            "sources: run tools/profile.sh + tools/aggregate_profile.py): HBM fraction only; the path is fp64-ALU / latency bound"}
        out = {
            "metric": "worlds*timesteps/sec fwd+bwd", "value": value, "unit": "worlds*timesteps/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "reps": len(R["reps"]), "reps_ms_per_step": [round(r / args.steps * 1e3, 4) for r in R["reps"]], "timed_seconds": sum(R["reps"]),
            "config": {"workload": f"{wl_desc}; batch={B} worlds/GPU; fwd+bwd through the C ABI, cold LCP start each step" +
                                   ("; ONE model handle in deferred-join mode (its slices run on the handle's own streams, the per-slice loss on them)" if one else f"; {len(bounds)} World handle(s), one stream each") +
                                   (f"; one step = one {args.rollout}-step rollout fwd+bwd (warm-started after its first step)" if args.rollout else ""),
                       "n_dofs": n, "contacts": m_rows // 3, "lcp_rows": m_rows, "worlds_per_gpu": B, "dt": md.dt,
                       "joint_noise": args.joint_noise if has_contact else None, "rollout_T": args.rollout or None, "rollout_checkpoint_every": (args.checkpoint_every or None) if args.rollout else None,
                       "saved_record_bytes_per_world_step": dims["saved_bytes_per_world"],
                       "rccl_world_size": (dist.get_world_size() if use_dist else 0),
                       # per-rank spread of the reported (median) repetition: the slowest rank is the one that counts (ms_per_step), the
                       # fastest says how far apart the ranks are - so that the first real multi-GPU run explains itself
                       "ranks_ms_per_step": ({"max": elapsed / args.steps * 1e3,
                                              "min": R["rank_min"][R["reps"].index(elapsed)] / args.steps * 1e3} if use_dist else None),
                       "devices_visible": torch.cuda.device_count(), "device_of_rank0": torch.cuda.current_device(),
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "lanes_with_contact": float((st & 0x1).astype(bool).mean()),
                       "lanes_resolved_at_lcp_stage0": float((st & 0x2).astype(bool).mean()) if m_rows else None,
                       "lanes_unresolved": float((st & 0x20).astype(bool).mean()),
                       "collective": "1 all-gather of the shared-control gradient per timed region"},
            "roofline": roof,
        }
        if easy is not None:
            est = easy["status"]
            esteps = max(1, args.steps // 2)
            out["secondary"] = {"stage0_only": {
                "joint_noise": args.easy_noise, "value": units_per_step * esteps / easy["elapsed"], "unit": "worlds*timesteps/s",
                "steps": esteps, "ms_per_step": easy["elapsed"] / esteps * 1e3,
                "lanes_resolved_at_lcp_stage0": float((est & 0x2).astype(bool).mean()),
                "note": "same workload on the easy pose distribution (every world short-circuits at LCP stage 0): round 1's headline regime"}}
        if single is not None:
            ssteps = max(1, args.steps // 2)
            out.setdefault("secondary", {})["single_stream"] = {
                "value": units_per_step * ssteps / single["elapsed"], "unit": "worlds*timesteps/s", "steps": ssteps,
                "ms_per_step": single["elapsed"] / ssteps * 1e3, "stream_slices": 1,
                "note": "the same batch and distribution with ONE launch per kernel per step (all worlds of the GPU in one chain, no stream slices)"}
        if single_call is not None:
            ssteps = max(1, args.steps // 2)
            out.setdefault("secondary", {})["single_call"] = {
                "value": units_per_step * ssteps / single_call["elapsed"], "unit": "worlds*timesteps/s", "steps": ssteps,
                "ms_per_step": single_call["elapsed"] / ssteps * 1e3, "stream_slices": single_call["slices"],
                "note": "the same batch and distribution handed to ONE World in one JOINED call per pass - what timestep() gives behind autograd: the "
                        "library slices the call itself and every call joins before it returns"}
        if one_handle_deferred is not None:
            out.setdefault("secondary", {})["four_handles"] = {
                "value": units_per_step * args.steps / one_handle_deferred["elapsed"], "unit": "worlds*timesteps/s", "steps": args.steps,
                "ms_per_step": one_handle_deferred["elapsed"] / args.steps * 1e3, "handles": len(bounds),
                "note": "round 5's harness pattern: the batch cut into one World handle per slice by the caller, every handle on its own HIP stream "
                        "(what `value` - ONE handle in deferred-join mode, nbl_set_deferred_join - replaces)"}
        if host_tensors is not None:
            out.setdefault("secondary", {})["host_tensors"] = {
                "value": B * host_tensors["steps"] / host_tensors["elapsed"], "unit": "worlds*timesteps/s", "steps": host_tensors["steps"],
                "ms_per_step": host_tensors["elapsed"] / host_tensors["steps"] * 1e3,
                "note": "PCIe-inclusive: timestep(world, state, action) + backward() of ONE World with float64 CPU tensors [B, 2n] / [B, k] in and CPU tensors "
                        "out (the reference's own convention); the World's pinned staging buffers + asynchronous copies on the step's stream; torch CPU "
                        "intra-op threads = 1 for this leg (the default overruns the container's CFS quota, see bench.py).  Never `value`.",
                "torch_cpu_threads": 1}
        if world_size == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(md, s_np, a_np)
            except Exception as e:  # the checker must never take the bench down
                out["cpu_baseline"] = {"value": None, "unit": "worlds*timesteps/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
