"""Rank-k update of a resident state against the pseudo-observation path (GPU box): python tools/update_bench.py (BLR_MI355X_SWEEP=always for the sweep column)
Prints us per call (B = 1) and updates/s (B = 2048) at D = 128 / 64 for k = 1..64, f64 -- the numbers quoted in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import blr_amd  # noqa
from blr_amd import _abi as a

dev = torch.device("cuda:0")
h = a.Handle(0)
h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
h.set_async(True)
dt, ndt = torch.float64, np.float64


def bench(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for D in (128, 64):
    for nb in (1, 2048):
        print(f"D={D} B={nb}")
        g = torch.Generator(device=dev).manual_seed(1)
        U = torch.triu(torch.randn((nb, D, D), generator=g, dtype=dt, device=dev)) * (0.3 / np.sqrt(D))
        U = U + torch.diag_embed(1.0 + U.diagonal(dim1=1, dim2=2).abs())
        T0 = U.transpose(1, 2).contiguous()  # column-major upper factor
        for k in (1, 3, 8, 16, 32, 64):
            X = torch.randn((nb, k, D), generator=g, dtype=dt, device=dev)
            y = torch.randn((nb, k), generator=g, dtype=dt, device=dev)
            s = torch.full((1,), 0.5, dtype=dt, device=dev)
            mw = torch.zeros((nb, D), dtype=dt, device=dev)
            T = T0.clone()
            mo, To = torch.empty_like(mw), torch.empty_like(T)
            lp = torch.zeros(nb, dtype=torch.float64, device=dev)
            info = torch.zeros(nb, dtype=torch.int32, device=dev)

            def upd():
                h.update_factor(ndt, a.MEM_DEVICE, a.LAYOUT_COLVECS, nb, D, k, X.data_ptr(), D, k * D, y.data_ptr(), k, a.NOISE_ISOTROPIC,
                                s.data_ptr(), 0, mw.data_ptr(), D, T.data_ptr(), D, D * D, lp.data_ptr(), info.data_ptr())

            def pseudo():
                h.posterior_batched(ndt, a.MEM_DEVICE, a.LAYOUT_COLVECS, nb, D, k, X.data_ptr(), D, k * D, y.data_ptr(), k, a.NOISE_ISOTROPIC,
                                    s.data_ptr(), 0, a.PRIOR_UPPER_FACTOR, mw.data_ptr(), D, T0.data_ptr(), D, D * D, mo.data_ptr(), D,
                                    To.data_ptr(), D, D * D, None, D, D * D, lp.data_ptr(), info.data_ptr())

            reps = 200 if nb == 1 else 20
            tu, tp = bench(upd, reps), bench(pseudo, reps)
            route = "sweep" if (k <= 16 and os.environ.get("BLR_MI355X_SWEEP") == "always") else "auto"
            print(f"  k={k:3d} {route:8s} update_factor {1e6 * tu:9.1f} us  ({nb / tu:12.0f} /s)   pseudo-observation path {1e6 * tp:9.1f} us  "
                  f"({nb / tp:12.0f} /s)   ratio {tp / tu:5.2f}")
