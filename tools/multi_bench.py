"""Timing of the shared-X multi-output evidence (blr_logpdf_multi_*) against S independent fused updates on the same X."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blr_amd
from blr_amd import _abi


def run(D, N, S, dtype):
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    X = torch.randn((N, D), generator=g, dtype=tdt, device=dev)  # D x N column-major
    Y = torch.randn((S, N), generator=g, dtype=tdt, device=dev)  # N x S column-major
    s = torch.full((1,), 0.1, dtype=tdt, device=dev)
    mw = torch.zeros((D,), dtype=tdt, device=dev)
    d = torch.ones((D,), dtype=tdt, device=dev)
    lp = torch.zeros(S, dtype=torch.float64, device=dev); lp2 = torch.zeros(S, dtype=torch.float64, device=dev)
    info = torch.zeros(S, dtype=torch.int32, device=dev)
    h = _abi.default_handle()

    def multi():
        h.logpdf_multi(dtype, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, D, N, S, X.data_ptr(), D, Y.data_ptr(), N, _abi.NOISE_ISOTROPIC,
                       s.data_ptr(), _abi.PRIOR_DIAGONAL, mw.data_ptr(), d.data_ptr(), 1, lp.data_ptr(), None, D, info.data_ptr())

    def batched():
        h.posterior_batched(dtype, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, S, D, N, X.data_ptr(), D, 0, Y.data_ptr(), N,
                            _abi.NOISE_ISOTROPIC, s.data_ptr(), 0, _abi.PRIOR_DIAGONAL, mw.data_ptr(), 0, d.data_ptr(), 1, 0,
                            None, D, None, D, D * D, None, D, D * D, lp2.data_ptr(), info.data_ptr())

    out = []
    for fn in (multi, batched):
        for _ in range(2):
            fn()
        h.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            fn()
        h.synchronize(); out.append((time.perf_counter() - t0) / 5)
    err = float(((lp - lp2).abs() / lp2.abs()).max())
    print(f"D={D} N={N} S={S} {np.dtype(dtype).name}: shared-X {out[0]*1e3:.3f} ms, {S} independent updates {out[1]*1e3:.3f} ms "
          f"({out[1]/out[0]:.1f}x), max rel diff {err:.1e}")


if __name__ == "__main__":
    run(128, 4096, 64, np.float64); run(128, 4096, 1024, np.float64); run(1024, 65536, 64, np.float32); run(2048, 16384, 32, np.float32)
