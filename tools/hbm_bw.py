"""HBM streaming bandwidth sanity check (torch copy / read-reduce) -- tells a slow box from a slow kernel."""
import torch, time
x = torch.empty(2 * 1024**3 // 4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
x.normal_()
for name, fn, bytes_ in (("copy", lambda: y.copy_(x), 2 * x.numel() * 4), ("sum", lambda: x.sum(), x.numel() * 4)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {bytes_ / dt / 1e12:.2f} TB/s")
