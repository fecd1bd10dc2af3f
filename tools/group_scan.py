"""Batches at D > 128: marginals and the evidence gradient with the regressors sharing every launch (default) against one regressor
at a time (CHAIN_BATCH=1, set per call through blr_set_option).  VERDICT r3 #6.   python tools/group_scan.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blr_amd
from blr_amd import _abi as a
dev = torch.device("cuda:0"); h = a.Handle(0); h.set_stream(torch.cuda.current_stream(dev).cuda_stream); h.set_async(True)


def timeit(fn, steps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


for dname, dt, ndt in (("f64", torch.float64, np.float64), ("f32", torch.float32, np.float32)):
    for B, D, N in ((32, 256, 4096), (17, 512, 8192)):
        g = torch.Generator(device=dev).manual_seed(3)
        X = torch.randn((B, N, D), generator=g, dtype=dt, device=dev); y = torch.randn((B, N), generator=g, dtype=dt, device=dev)
        s = torch.exp(0.3 * torch.randn((B, N), generator=g, dtype=dt, device=dev)); mw = torch.randn((B, D), generator=g, dtype=dt, device=dev)
        d = torch.ones((B, D), dtype=dt, device=dev)
        lp = torch.zeros(B, dtype=torch.float64, device=dev); info = torch.zeros(B, dtype=torch.int32, device=dev)
        dX = torch.empty_like(X); dy = torch.empty_like(y); ds = torch.empty_like(s); dmw = torch.empty_like(mw); mwp = torch.empty_like(mw)
        U = torch.eye(D, dtype=dt, device=dev).repeat(B, 1, 1) * 1.5; mean = torch.empty_like(y); var = torch.empty_like(y)

        def grad():
            h.logpdf_grad_batched(ndt, a.MEM_DEVICE, a.LAYOUT_COLVECS, B, D, N, X.data_ptr(), D, N * D, y.data_ptr(), N, a.NOISE_DIAGONAL, s.data_ptr(), N,
                                  a.PRIOR_DIAGONAL, mw.data_ptr(), D, d.data_ptr(), 1, D, lp.data_ptr(), dX.data_ptr(), D, N * D, dy.data_ptr(), N, ds.data_ptr(), N,
                                  dmw.data_ptr(), D, mwp.data_ptr(), D, None, D, D * D, info.data_ptr())

        def marg():
            h.marginals_batched(ndt, a.MEM_DEVICE, a.LAYOUT_COLVECS, B, D, N, X.data_ptr(), D, N * D, a.NOISE_DIAGONAL, s.data_ptr(), N, a.PRIOR_UPPER_FACTOR,
                                mw.data_ptr(), D, U.data_ptr(), D, D * D, mean.data_ptr(), N, var.data_ptr(), N, info.data_ptr())

        for name, fn in (("logpdf_grad_batched", grad), ("marginals_batched", marg)):
            t_g = timeit(fn)
            assert int(info.abs().sum().item()) == 0
            h.set_option("CHAIN_BATCH", "1")
            t_1 = timeit(fn, 3)
            h.set_option("CHAIN_BATCH", None)
            print(f"{name} {dname} B={B} D={D} N={N}: grouped {1e3*t_g:.3f} ms, one regressor at a time {1e3*t_1:.3f} ms ({t_1/t_g:.1f}x)")
