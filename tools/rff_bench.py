"""BASELINE config 5 end to end: random-Fourier features on the device + fused posterior/logpdf (blr_posterior_rff_f32)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blr_amd
from blr_amd import _abi


def run(Din, D, N, dtype):
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    Xin = torch.randn((N, Din), generator=g, dtype=tdt, device=dev)     # Din x N column-major
    Om = torch.randn((D, Din), generator=g, dtype=tdt, device=dev)      # Din x D column-major
    ph = (2 * np.pi) * torch.rand((D,), generator=g, dtype=tdt, device=dev)
    y = torch.randn((N,), generator=g, dtype=tdt, device=dev)
    s = torch.full((1,), 0.1, dtype=tdt, device=dev)
    mw = torch.zeros((D,), dtype=tdt, device=dev); d = torch.ones((D,), dtype=tdt, device=dev)
    mwp = torch.empty((D,), dtype=tdt, device=dev); Tp = torch.empty((D, D), dtype=tdt, device=dev)
    lp = torch.zeros(1, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
    h = _abi.default_handle()
    h.set_async(True)

    def call():
        h.posterior_rff(dtype, _abi.MEM_DEVICE, Din, D, N, Xin.data_ptr(), Din, Om.data_ptr(), Din, ph.data_ptr(), float(np.sqrt(2.0 / D)),
                        y.data_ptr(), _abi.NOISE_ISOTROPIC, s.data_ptr(), _abi.PRIOR_DIAGONAL, mw.data_ptr(), d.data_ptr(), 1,
                        mwp.data_ptr(), Tp.data_ptr(), D, None, D, lp.data_ptr(), info.data_ptr())

    for _ in range(3):
        call()
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        call()
    h.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"RFF Din={Din} -> D={D}, N={N} {np.dtype(dtype).name}: {dt*1e3:.3f} ms per update ({1/dt:.0f} updates/s), logpdf {float(lp):.3f}")


if __name__ == "__main__":
    run(8, 2048, 16384, np.float32); run(8, 2048, 16384, np.float64); run(8, 128, 4096, np.float64)
