"""Practical HBM ceilings on this GPU: device-to-device copy, fill, read-only sum (GB/s of bytes moved)."""
import torch, time
dev = torch.device("cuda", 0)
n = 256 * 1024 * 1024   # 1 GiB of float32
a = torch.empty(n, device=dev); b = torch.randn(n, device=dev)
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
t = timeit(lambda: a.copy_(b)); print(f"copy   {2*n*4/t/1e9:8.0f} GB/s")
t = timeit(lambda: a.fill_(1.0)); print(f"fill   {n*4/t/1e9:8.0f} GB/s")
t = timeit(lambda: b.sum()); print(f"read   {n*4/t/1e9:8.0f} GB/s")
t = timeit(lambda: torch.add(a, b, out=a)); print(f"a+=b   {3*n*4/t/1e9:8.0f} GB/s")
