"""RowVecs against ColVecs storage on the two product-form kernels at D = 128 (marginals, evidence gradient): python tools/rowvecs_gap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blr_amd
from blr_amd import _abi as a
dev = torch.device("cuda:0"); h = a.Handle(0); h.set_stream(torch.cuda.current_stream(dev).cuda_stream); h.set_async(True)
dt, ndt = torch.float64, np.float64
def timeit(fn, steps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps
g = torch.Generator(device=dev).manual_seed(3)
B, D, N = 64, 128, 4096
X = torch.randn((B, N, D), generator=g, dtype=dt, device=dev); Xr = X.transpose(1, 2).contiguous()
s = torch.full((1,), 0.1, dtype=dt, device=dev); mw = torch.randn((B, D), generator=g, dtype=dt, device=dev)
U = (torch.eye(D, dtype=dt, device=dev) * 1.5).repeat(B, 1, 1); mean = torch.empty((B, N), dtype=dt, device=dev); var = torch.empty_like(mean); info = torch.zeros(B, dtype=torch.int32, device=dev)
for name, lay, Xp, ldx in (("ColVecs", a.LAYOUT_COLVECS, X, D), ("RowVecs", a.LAYOUT_ROWVECS, Xr, N)):
    t = timeit(lambda: h.marginals_batched(ndt, a.MEM_DEVICE, lay, B, D, N, Xp.data_ptr(), ldx, N * D, a.NOISE_ISOTROPIC, s.data_ptr(), 0, a.PRIOR_UPPER_FACTOR, mw.data_ptr(), D, U.data_ptr(), D, D * D, mean.data_ptr(), N, var.data_ptr(), N, info.data_ptr()))
    print(f"marginals 64 x (128, 4096) f64 {name}: {1e3*t:.3f} ms")
B = 1024
X = torch.randn((B, N, D), generator=g, dtype=dt, device=dev); Xr = X.transpose(1, 2).contiguous(); y = torch.randn((B, N), generator=g, dtype=dt, device=dev)
mw = torch.randn((B, D), generator=g, dtype=dt, device=dev); d = torch.ones((D,), dtype=dt, device=dev); lp = torch.zeros(B, dtype=dt, device=dev); info = torch.zeros(B, dtype=torch.int32, device=dev)
dX = torch.empty_like(X); dy = torch.empty_like(y); ds = torch.empty_like(y); dmw = torch.empty_like(mw); mwp = torch.empty_like(mw); Ai = torch.empty((B, D, D), dtype=dt, device=dev)
for name, lay, Xp, ldx in (("ColVecs", a.LAYOUT_COLVECS, X, D), ("RowVecs", a.LAYOUT_ROWVECS, Xr, N)):
    t = timeit(lambda: h.logpdf_grad_batched(ndt, a.MEM_DEVICE, lay, B, D, N, Xp.data_ptr(), ldx, N * D, y.data_ptr(), N, a.NOISE_ISOTROPIC, s.data_ptr(), 0, a.PRIOR_DIAGONAL, mw.data_ptr(), D, d.data_ptr(), 1, 0,
                                             lp.data_ptr(), dX.data_ptr(), ldx, N * D, dy.data_ptr(), N, ds.data_ptr(), N, dmw.data_ptr(), D, mwp.data_ptr(), D, Ai.data_ptr(), D, D * D, info.data_ptr()), 5)
    print(f"gradient 1024 x (128, 4096) f64 {name}: {1e3*t:.3f} ms")
