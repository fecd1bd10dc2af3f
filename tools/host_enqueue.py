"""Host enqueue time vs device time of one large-D update (experiments): python tools/host_enqueue.py [c3|c5] [side]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import blr_amd  # noqa
from blr_amd import _abi
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda:0")
if len(sys.argv) > 2:
    torch.cuda.set_stream(torch.cuda.Stream(dev))
h = _abi.Handle(0)
h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
h.set_async(True)
D, N, noise, Din = (1024, 65536, "diagonal", None) if cfg == "c3" else (2048, 16384, "isotropic", 8)
wl = bench.Workload(torch, _abi, h, dev, cfg, 1, D, N, "f32", noise, 1, Din)
for _ in range(3):
    wl.launch()
torch.cuda.synchronize()
for reps in (1, 5, 20):
    t0 = time.perf_counter()
    for _ in range(reps):
        wl.launch()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{cfg} reps {reps}: host enqueue {1e3 * (t1 - t0) / reps:.3f} ms/step, total {1e3 * (t2 - t0) / reps:.3f} ms/step")
