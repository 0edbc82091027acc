"""Reference point for the Adam kernel (csrc/train_ops.hip: gsr_adam_step): one Adam step of the [2 M, 16, 3] SH tensor (96 M
floats: 7 x 4 bytes per element of compulsory traffic) with (a) gsr_adam_step, (b) torch.optim.Adam(fused=True) -- ATen's
multi-tensor fused kernel, (c) torch.optim.Adam(foreach=True), (d) the default per-op implementation.  Prints one JSON line."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.load_package()
from photo_slam_amd import capi, rasterize_points as rp  # noqa: E402

dev = torch.device("cuda", 0)
lib = capi.load()
n_rows = 2_000_000
p = torch.randn(n_rows, 16, 3, device=dev)
g = 1e-3 * torch.randn_like(p)
m, v = torch.zeros_like(p), torch.zeros_like(p)


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


step = [0]
def ours():
    step[0] += 1
    capi.check(lib, lib.gsr_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 0.0025, 0.9, 0.999, 1e-15,
                                      step[0], 0, 0, 0.0025, rp._stream_ptr(p)), "gsr_adam_step")

out = {"elements": p.numel(), "compulsory_MB": round(p.numel() * 28 / 1e6, 1)}
t = timeit(ours)
out["gsr_adam_step"] = {"ms": round(t, 4), "TBps": round(p.numel() * 28 / t / 1e9, 2)}
for name, kw in (("torch_fused", dict(fused=True)), ("torch_foreach", dict(foreach=True)), ("torch_default", dict(foreach=False, fused=False))):
    q = torch.nn.Parameter(p.clone())
    q.grad = g.clone()
    try:
        opt = torch.optim.Adam([q], lr=0.0025, eps=1e-15, **kw)
        t = timeit(opt.step)
        out[name] = {"ms": round(t, 4), "TBps": round(p.numel() * 28 / t / 1e9, 2)}
    except Exception as e:   # noqa: BLE001
        out[name] = {"error": repr(e)[:200]}
    del q
print(json.dumps(out))
