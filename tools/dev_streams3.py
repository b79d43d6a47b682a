import os, time, torch
print({k: v for k, v in os.environ.items() if any(t in k.upper() for t in ("HIP", "HSA", "GPU_", "ROCR", "AMD_"))})
dev = "cuda:0"
# a latency-bound tiny kernel: long dependent chain on a small tensor
x = [torch.randn(1024, device=dev, dtype=torch.float64) for _ in range(4)]
def work(t):
    for _ in range(200):
        t = torch.sin(t) * 1.0001
    return t
for ns in (1, 2, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    torch.cuda.synchronize()
    t0 = time.time()
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            work(x[i])
    torch.cuda.synchronize()
    print(ns, "streams:", (time.time() - t0) * 1e3, "ms")
