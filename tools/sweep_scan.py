import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import blr_amd
from blr_amd import _abi as a
dev = torch.device("cuda:0"); h = a.Handle(0); h.set_stream(torch.cuda.current_stream(dev).cuda_stream); h.set_async(True)
h.set_option("SWEEP", "always")
dt, ndt = torch.float64, np.float64
def bench(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
D = 128
for nb in (1, 256, 512, 1024, 2048):
    for k in (1, 4):
        g = torch.Generator(device=dev).manual_seed(1)
        U = torch.triu(torch.randn((nb, D, D), generator=g, dtype=dt, device=dev)) * (0.3 / np.sqrt(D))
        U = U + torch.diag_embed(1.0 + U.diagonal(dim1=1, dim2=2).abs())
        T = U.transpose(1, 2).contiguous()
        X = torch.randn((nb, k, D), generator=g, dtype=dt, device=dev); y = torch.randn((nb, k), generator=g, dtype=dt, device=dev)
        s = torch.full((1,), 0.5, dtype=dt, device=dev); mw = torch.zeros((nb, D), dtype=dt, device=dev)
        lp = torch.zeros(nb, dtype=torch.float64, device=dev); info = torch.zeros(nb, dtype=torch.int32, device=dev)
        def upd():
            h.update_factor(ndt, a.MEM_DEVICE, a.LAYOUT_COLVECS, nb, D, k, X.data_ptr(), D, k * D, y.data_ptr(), k, a.NOISE_ISOTROPIC,
                            s.data_ptr(), 0, mw.data_ptr(), D, T.data_ptr(), D, D * D, lp.data_ptr(), info.data_ptr())
        t = bench(upd, 50)
        print(f"B={nb:5d} k={k}: {1e6*t:8.1f} us per call")
