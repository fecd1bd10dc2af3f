"""Timing of value + gradient of the log marginal likelihood on device-resident inputs (blr_logpdf_grad_batched_*)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blr_amd
from blr_amd import _abi


def run(B, D, N, dtype):
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    X = torch.randn((B, N, D), generator=g, dtype=tdt, device=dev)
    y = torch.randn((B, N), generator=g, dtype=tdt, device=dev)
    s = torch.full((1,), 0.1, dtype=tdt, device=dev)
    mw = torch.zeros((B, D), dtype=tdt, device=dev)
    d = torch.ones((D,), dtype=tdt, device=dev)
    lp = torch.zeros(B, dtype=torch.float64, device=dev); info = torch.zeros(B, dtype=torch.int32, device=dev)
    dX = torch.empty_like(X); dy = torch.empty_like(y); ds = torch.empty_like(y); dmw = torch.empty_like(mw)
    mwp = torch.empty_like(mw); Ai = torch.empty((B, D, D), dtype=tdt, device=dev)
    h = _abi.default_handle()
    h.set_async(True)

    def call():
        h.logpdf_grad_batched(dtype, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, B, D, N, X.data_ptr(), D, N * D, y.data_ptr(), N,
                              _abi.NOISE_ISOTROPIC, s.data_ptr(), 0, _abi.PRIOR_DIAGONAL, mw.data_ptr(), D, d.data_ptr(), 1, 0,
                              lp.data_ptr(), dX.data_ptr(), D, N * D, dy.data_ptr(), N, ds.data_ptr(), N, dmw.data_ptr(), D,
                              mwp.data_ptr(), D, Ai.data_ptr(), D, D * D, info.data_ptr())

    for _ in range(3):
        call()
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        call()
    h.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"value+gradient B={B} D={D} N={N} {np.dtype(dtype).name}: {dt*1e3:.3f} ms  {B/dt/1e3:.1f} k evaluations/s")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "batch":  # batches at D > 128
        run(16, 256, 4096, np.float32); run(8, 512, 4096, np.float32); run(8, 1024, 16384, np.float32); run(8, 512, 4096, np.float64)
    elif len(sys.argv) > 1 and sys.argv[1] == "large":
        run(1, 1024, 65536, np.float32); run(1, 2048, 16384, np.float32); run(1, 1024, 65536, np.float64)
    else:
        run(1024, 128, 4096, np.float64); run(1024, 128, 4096, np.float32); run(8192, 64, 1024, np.float64)
