"""Timing of the batched marginal stream (mean + var) on device-resident inputs: B regressors x N inputs at dimension D."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blr_amd
from blr_amd import _abi

def run(B, D, N, dtype, mean_only=False):
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.randn((B, N, D), generator=g, dtype=tdt, device=dev)
    mw = torch.randn((B, D), generator=g, dtype=tdt, device=dev)
    U = torch.triu(torch.randn((B, D, D), generator=g, dtype=tdt, device=dev)) / D**0.5 + 2 * torch.eye(D, dtype=tdt, device=dev)
    Ucm = U.transpose(1, 2).contiguous()  # column-major storage of U
    s = torch.full((1,), 0.1, dtype=tdt, device=dev)
    mean = torch.empty((B, N), dtype=tdt, device=dev); var = torch.empty((B, N), dtype=tdt, device=dev)
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    h = _abi.default_handle()
    def call():
        h.marginals_batched(dtype, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, B, D, N, X.data_ptr(), D, N * D, _abi.NOISE_ISOTROPIC,
                            s.data_ptr(), 0, _abi.PRIOR_UPPER_FACTOR, mw.data_ptr(), D, Ucm.data_ptr(), D, D * D,
                            mean.data_ptr(), N, 0 if mean_only else var.data_ptr(), N, info.data_ptr())
    for _ in range(3): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    if mean_only:
        m_ref = (X[0].double() @ mw[0].double())
        err = ((mean[0].double() - m_ref).abs().max() / m_ref.abs().max()).item()
        bytes_ = B * N * D * X.element_size()
        print(f"mean only B={B} D={D} N={N} {np.dtype(dtype).name}: {dt*1e3:.3f} ms  {B*N/dt/1e6:.1f} M means/s  {bytes_/dt/1e12:.2f} TB/s  max err {err:.2e}")
        return
    # reference check on one regressor
    Xb = X[0].T.double(); Ub = U[0].double()
    alpha = torch.linalg.solve_triangular(Ub.T, Xb, upper=False)
    v_ref = (alpha * alpha).sum(0) + 0.1
    err = ((var[0].double() - v_ref).abs() / v_ref).max().item()
    print(f"B={B} D={D} N={N} {np.dtype(dtype).name}: {dt*1e3:.3f} ms  {B*N/dt/1e6:.1f} M marginals/s  {B*N*D*D/dt/1e12:.2f} TFLOP/s  max rel err {err:.2e}")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "large":
        run(1, 1024, 65536, np.float32); run(1, 2048, 16384, np.float32); run(1, 1024, 65536, np.float64)
    else:
        run(64, 128, 4096, np.float64); run(64, 128, 4096, np.float32); run(512, 64, 1024, np.float64)
        run(64, 128, 4096, np.float64, True); run(1024, 128, 4096, np.float64, True); run(1, 1024, 65536, np.float32, True)
