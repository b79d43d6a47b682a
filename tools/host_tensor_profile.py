"""Where the time of a host-tensor step goes (timestep() + backward() with CPU float64 tensors in and out, one World, B = 4096):
every phase timed with a device synchronisation after it.  Developer aid behind DESIGN.md's PCIe-inclusive figures."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import nimblephysics_amd as na
from nimblephysics_amd.timestep import timestep
from util import contact_inputs

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
md, s, a = contact_inputs("atlas20", B, 1000, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
w = na.World(md, device=dev)
S, A = torch.tensor(s), torch.tensor(a)


def sync():
    torch.cuda.synchronize()


def T(f, reps=20):
    for _ in range(3):
        f()
    sync(); t0 = time.perf_counter()
    for _ in range(reps):
        f()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3


def whole():
    st = S.clone().requires_grad_(True); at = A.clone().requires_grad_(True)
    w.reset_lcp_cache()
    out = timestep(w, st, at)
    out.backward(2.0 * out.detach())
    return st.grad, at.grad


spans = []


def timed(obj, name):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); sync(); spans.append((name, (time.perf_counter() - t0) * 1e3)); return r
    setattr(obj, name, g)


if os.environ.get("SPANS"):
    for nm in ("_prep", "_to_host", "step_soa", "backward_soa", "to_soa", "from_soa"):
        timed(w, nm)
    for i in range(60):
        spans.clear(); sync(); t0 = time.perf_counter()
        st = S.clone().requires_grad_(True); at = A.clone().requires_grad_(True)
        t1 = time.perf_counter(); w.reset_lcp_cache(); out = timestep(w, st, at); sync(); t2 = time.perf_counter()
        gg = 2.0 * out.detach(); t3 = time.perf_counter(); out.backward(gg); sync(); t4 = time.perf_counter()
        del st, at, out, gg; t5 = time.perf_counter()
        if t5 - t0 > 5e-3:
            print(i, "clone %.2f fwd %.2f mul %.2f bwd %.2f del %.2f" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)),
                  " ".join("%s=%.2f" % s_ for s_ in spans))
    sys.exit(0)
its = []
for i in range(40):
    sync(); t0 = time.perf_counter(); whole(); sync(); its.append((time.perf_counter() - t0) * 1e3)
print("per-iteration ms of the whole host-tensor step:", " ".join("%.2f" % x for x in its))
print("whole host-tensor step: %.3f ms" % T(whole))
print("  clone+requires_grad: %.3f ms" % T(lambda: (S.clone().requires_grad_(True), A.clone().requires_grad_(True))))
print("  pinned empty [B,2n]: %.3f ms" % T(lambda: torch.empty(S.shape, dtype=torch.float64, pin_memory=True)))
pin = torch.empty(S.shape, dtype=torch.float64, pin_memory=True)
print("  pageable -> pinned copy [B,2n]: %.3f ms" % T(lambda: pin.copy_(S)))
print("  pinned -> device non_blocking [B,2n]: %.3f ms" % T(lambda: pin.to(dev, non_blocking=True)))
print("  pageable -> device (.to) [B,2n]: %.3f ms" % T(lambda: S.to(dev)))
print("  _prep(state): %.3f ms" % T(lambda: w._prep(S, 2 * w.n, "x")))
d = S.to(dev)
print("  to_soa: %.3f ms" % T(lambda: w.to_soa(d)))
print("  _to_host(one [B,2n]): %.3f ms" % T(lambda: w._to_host(d)))
print("  device -> pageable (.cpu()): %.3f ms" % T(lambda: d.cpu()))
sd, ad = S.to(dev), A.to(dev)


def devstep():
    st = sd.clone().requires_grad_(True); at = ad.clone().requires_grad_(True)
    w.reset_lcp_cache()
    out = timestep(w, st, at)
    out.backward(2.0 * out.detach())


print("  device-tensor step (same World): %.3f ms" % T(devstep))
print("  2.0 * out on the host [B,2n]: %.3f ms" % T(lambda: 2.0 * S))
try:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(5):
            whole()
        sync()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=18))
except Exception as e:
    print("profiler unavailable:", repr(e)[:200])
