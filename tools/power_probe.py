"""Is a Gram launch bound by the clock the part holds (DVFS) rather than by its instruction stream?  Same kernel, same shapes,
operands zero / constant / random: only the toggling of the datapath differs.  (GPU box)
    python tools/power_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blr_amd
from blr_amd import _abi

dev = torch.device("cuda:0")
h = _abi.Handle(0)
stream = torch.cuda.current_stream(dev)
h.set_stream(stream.cuda_stream); h.set_async(True)


def run(name, B, D, N, dt, fill, diag):
    tdt = torch.float64 if dt == np.float64 else torch.float32
    if fill == "random":
        X = torch.randn((B, N, D), dtype=tdt, device=dev)
    elif fill == "zeros":
        X = torch.zeros((B, N, D), dtype=tdt, device=dev)
    else:
        X = torch.full((B, N, D), 0.5, dtype=tdt, device=dev)
    y = torch.randn((B, N), dtype=tdt, device=dev)
    s = torch.exp(torch.randn((B, N), dtype=tdt, device=dev)) if diag else torch.full((1,), 0.1, dtype=tdt, device=dev)
    mw = torch.zeros((B, D), dtype=tdt, device=dev); dpr = torch.ones((D,), dtype=tdt, device=dev)
    mwp = torch.empty((B, D), dtype=tdt, device=dev); Tp = torch.empty((B, D, D), dtype=tdt, device=dev)
    lp = torch.empty((B,), dtype=torch.float64, device=dev); info = torch.empty((B,), dtype=torch.int32, device=dev)
    a = _abi
    def call():
        h.posterior_batched(dt, a.MEM_DEVICE, a.LAYOUT_COLVECS, B, D, N, X.data_ptr(), D, N * D, y.data_ptr(), N,
                            a.NOISE_DIAGONAL if diag else a.NOISE_ISOTROPIC, s.data_ptr(), N if diag else 0, a.PRIOR_DIAGONAL, mw.data_ptr(), D,
                            dpr.data_ptr(), 1, 0, mwp.data_ptr(), D, Tp.data_ptr(), D, D * D, None, D, D * D, lp.data_ptr(), info.data_ptr())
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10): call()
    e1.record(stream); torch.cuda.synchronize()
    print(f"{name:28s} {fill:8s}: {e0.elapsed_time(e1) / 10:8.3f} ms per step")


for fill in ("zeros", "constant", "random"):
    run("c2 f64 B=4096 D=128 N=4096", 4096, 128, 4096, np.float64, fill, False)
for fill in ("zeros", "constant", "random"):
    run("c2 f32 B=4096 D=128 N=4096", 4096, 128, 4096, np.float32, fill, False)
for fill in ("zeros", "constant", "random"):
    run("c3 f32 D=1024 N=65536", 1, 1024, 65536, np.float32, fill, True)
for fill in ("zeros", "constant", "random"):
    run("c4 f64 B=8192 D=64 N=1024", 8192, 64, 1024, np.float64, fill, False)
