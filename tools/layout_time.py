"""c2's shape (4096 x (D = 128, N = 4096), fp64, isotropic noise, diagonal prior) in both layouts: python tools/layout_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blr_amd
from blr_amd import _abi as a
dev = torch.device("cuda:0"); h = a.Handle(0); h.set_stream(torch.cuda.current_stream(dev).cuda_stream); h.set_async(True)
dt, ndt = torch.float64, np.float64
B, D, N = 4096, 128, 4096
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((B, N, D), generator=g, dtype=dt, device=dev); y = torch.randn((B, N), generator=g, dtype=dt, device=dev)
s = torch.full((1,), 0.1, dtype=dt, device=dev); mw = torch.zeros((B, D), dtype=dt, device=dev); d1 = torch.ones((D,), dtype=dt, device=dev)
mo = torch.empty((B, D), dtype=dt, device=dev); To = torch.empty((B, D, D), dtype=dt, device=dev); lp = torch.zeros(B, dtype=torch.float64, device=dev); info = torch.zeros(B, dtype=torch.int32, device=dev)
for name, layout, ldx in (("ColVecs (D x N, column-major)", a.LAYOUT_COLVECS, D), ("RowVecs (N x D, column-major)", a.LAYOUT_ROWVECS, N)):
    def run():
        h.posterior_batched(ndt, a.MEM_DEVICE, layout, B, D, N, X.data_ptr(), ldx, N * D, y.data_ptr(), N, a.NOISE_ISOTROPIC, s.data_ptr(), 0, a.PRIOR_DIAGONAL, mw.data_ptr(), D,
                            d1.data_ptr(), 1, 0, mo.data_ptr(), D, To.data_ptr(), D, D * D, None, D, D * D, lp.data_ptr(), info.data_ptr())
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"{name}: {1e3*t:.3f} ms = {B/t/1e6:.3f} M updates/s")
