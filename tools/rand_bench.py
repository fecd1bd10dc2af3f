"""Timing of rand(rng, fx, S) with the normals already on the device (blr_rand_*): Y = X'(mw + U^-1 Z1) + sqrt(s) .* Z2."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blr_amd
from blr_amd import _abi


def run(D, N, S, dtype):
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(9)
    X = torch.randn((N, D), generator=g, dtype=tdt, device=dev)
    Z1 = torch.randn((S, D), generator=g, dtype=tdt, device=dev)
    Z2 = torch.randn((S, N), generator=g, dtype=tdt, device=dev)
    Y = torch.empty((S, N), dtype=tdt, device=dev)
    s = torch.full((1,), 0.1, dtype=tdt, device=dev)
    mw = torch.zeros((D,), dtype=tdt, device=dev)
    U = torch.triu(torch.randn((D, D), generator=g, dtype=tdt, device=dev)) / D**0.5 + 2 * torch.eye(D, dtype=tdt, device=dev)
    Ucm = U.T.contiguous()
    h = _abi.default_handle()

    def call():
        h.rand(dtype, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, D, N, S, X.data_ptr(), D, _abi.NOISE_ISOTROPIC, s.data_ptr(),
               _abi.PRIOR_UPPER_FACTOR, mw.data_ptr(), Ucm.data_ptr(), D, Z1.data_ptr(), D, Z2.data_ptr(), N, Y.data_ptr(), N)

    for _ in range(2):
        call()
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        call()
    h.synchronize(); dt = (time.perf_counter() - t0) / 5
    W = torch.linalg.solve_triangular(U.double(), Z1.double().T, upper=True)
    ref = X.double() @ W + (0.1 ** 0.5) * Z2.double().T
    err = float(((Y.double().T - ref).abs().max()) / ref.abs().max())
    print(f"rand D={D} N={N} S={S} {np.dtype(dtype).name}: {dt*1e3:.3f} ms  {2*D*N*S/dt/1e12:.2f} TFLOP/s  {N*S/dt/1e9:.2f} G outputs/s  max rel err {err:.1e}")


if __name__ == "__main__":
    run(128, 4096, 64, np.float64); run(128, 4096, 1024, np.float64); run(1024, 65536, 64, np.float32); run(128, 4096, 1024, np.float32)
