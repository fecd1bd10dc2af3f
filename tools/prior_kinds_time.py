import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import blr_amd
from blr_amd import _abi as a
dev = torch.device("cuda:0"); h = a.Handle(0); h.set_stream(torch.cuda.current_stream(dev).cuda_stream); h.set_async(True)
dt, ndt = torch.float64, np.float64
B, D, N = 4096, 128, 4096
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((B, N, D), generator=g, dtype=dt, device=dev); y = torch.randn((B, N), generator=g, dtype=dt, device=dev)
s = torch.full((1,), 0.1, dtype=dt, device=dev); mw = torch.zeros((B, D), dtype=dt, device=dev)
U = torch.triu(torch.randn((1, D, D), generator=g, dtype=dt, device=dev)) * (0.3 / np.sqrt(D)); U = U + torch.diag_embed(1.0 + U.diagonal(dim1=1, dim2=2).abs())
T0 = U.transpose(1, 2).contiguous()
d1 = torch.ones((D,), dtype=dt, device=dev)
mo = torch.empty((B, D), dtype=dt, device=dev); To = torch.empty((B, D, D), dtype=dt, device=dev); lp = torch.zeros(B, dtype=torch.float64, device=dev); info = torch.zeros(B, dtype=torch.int32, device=dev)
def run(pk, L, ldl, strideL):
    h.posterior_batched(ndt, a.MEM_DEVICE, a.LAYOUT_COLVECS, B, D, N, X.data_ptr(), D, N * D, y.data_ptr(), N, a.NOISE_ISOTROPIC, s.data_ptr(), 0, pk, mw.data_ptr(), D,
                        L.data_ptr(), ldl, strideL, mo.data_ptr(), D, To.data_ptr(), D, D * D, None, D, D * D, lp.data_ptr(), info.data_ptr())
for name, args in (("diagonal prior", (a.PRIOR_DIAGONAL, d1, 1, 0)), ("factor prior", (a.PRIOR_UPPER_FACTOR, T0, D, 0)), ("dense prior", (a.PRIOR_DENSE, (T0[0].T @ T0[0]).contiguous(), D, 0))):
    for _ in range(3): run(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run(*args)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"{name}: {1e3*t:.3f} ms = {B/t/1e6:.3f} M updates/s")
