#!/usr/bin/env python3
"""bench.py -- posterior-updates/s of the fused MI355X posterior+logpdf path.

Workload (BASELINE.json configs[1], "c2"): independent regressors at D=128, N=4096, ColVecs, isotropic
noise, fp64, Lw = I, inputs resident in HBM.  One *step* = one pass of the hot path over one batch of
`--batch` regressors per GPU: ONE launch of the fused kernel (Gram + Cholesky + solves + evidence,
producing mw', T and logpdf for every regressor), the fixed-order device sum of the batch's log evidences,
and -- for N > 1 ranks -- the single all-gather of the per-rank partial sums (the only collective).
One "posterior update" = one regressor's full (mw', T, logpdf).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# spec-sheet peaks: HBM3E 8.0 TB/s; fp64 matrix 78.6 TF; fp32 matrix 157.3 TF
PEAK_HBM_GBS = 8000.0
PEAK_TF = {"f64": 78.6, "f32": 157.3}
# what tools/mfma_peak.hip sustains on this pool: v_mfma_f64_16x16x4 tops out at 47.7 TF
MEASURED_MFMA_TF = {"f64": 47.7, "f32": 155.0}
# HBM bytes per regressor from the PMC passes summarised in profiles/r01_pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE)
PMC_TRAFFIC_BYTES_PER_UPDATE = {("f64", 128, 4096, "isotropic"): 19772626176.0 / 4096}


def algorithmic_bytes(D, N, w, diag_noise):
    """w(D N + N + N_s + 2D + 2D^2) + 8 per update."""
    ns = N if diag_noise else 1
    return w * (D * N + N + ns + 2 * D + 2 * D * D) + 8


def algorithmic_flops(D, N):
    """D(D+1)N (symmetric-half SYRK) + 4DN + D^3/3 + 3D^2 + 5N per update."""
    return D * (D + 1) * N + 4 * D * N + D**3 / 3 + 3 * D * D + 5 * N


def cpu_baseline(D, N, seconds, seed):
    """Reference algorithm restated (oracle: literal op sequence of reference :72-89 + :55-69 on OpenBLAS),
    independent regressors spread over the host cores with one BLAS thread each (the CPU analogue of the
    batched GPU launch).  Child processes are started BEFORE this process touches the GPU."""
    import subprocess

    workers = max(1, min((os.cpu_count() or 2) // 2, 64))
    script = os.path.join(ROOT, "oracle", "cpu_baseline_worker.py")
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, script, str(D), str(N), str(seconds), str(seed + i)],
                              stdout=subprocess.PIPE, text=True) for i in range(workers)]
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=seconds * 20 + 300)
        if p.returncode == 0 and out.strip():
            d, t = out.split()
            res.append((int(d), float(t)))
    wall = time.perf_counter() - t0
    if not res:
        return None
    rate = sum(d / t for d, t in res)
    single = rate / len(res)
    return {
        "value": rate,
        "unit": "posterior-updates/s",
        "cores": len(res),
        "kind": "port",
        "sample": f"{sum(d for d, _ in res)} regressors at D={D}, N={N}, fp64: oracle (NumPy/SciPy on OpenBLAS) running the "
                  f"reference's literal op sequence logpdf+posterior, {len(res)} worker processes x 1 BLAS thread for "
                  f"{seconds:.0f} s each ({single:.1f} updates/s per core; {os.cpu_count()} hardware threads on the host; "
                  f"{wall:.0f} s wall incl. start-up)",
    }


def _config_name(D, N, dtype):
    """BASELINE.json config the shape corresponds to (c2 is the headline; the others are secondary shapes)."""
    table = {(128, 4096, "f64"): "c2", (2, 10, "f64"): "c1", (1024, 65536, "f32"): "c3", (64, 1024, "f64"): "c4 shape",
             (2048, 16384, "f32"): "c5 shape (features precomputed)"}
    return table.get((D, N, dtype), "custom")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="regressors per GPU per step (17 GB of X at the c2 shape)")
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--noise", choices=["isotropic", "diagonal"], default="isotropic")
    ap.add_argument("--cpu-seconds", type=float, default=None,
                    help="budget of the cpu_baseline leg (0 = skip; default: 10 s at the headline shape, skipped otherwise)")
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default=None,
                    help="BASELINE.json shape presets: c2 = the headline workload (default); c3/c4/c5 = the secondary shapes "
                         "(c5: features precomputed -- the end-to-end RFF call is timed by tools/rff_bench.py)")
    args = ap.parse_args()
    if args.config == "c3":
        args.D, args.N, args.dtype, args.noise, args.batch = 1024, 65536, "f32", "diagonal", 1
    elif args.config == "c4":
        args.D, args.N, args.dtype, args.noise, args.batch = 64, 1024, "f64", "isotropic", 8192
    elif args.config == "c5":
        args.D, args.N, args.dtype, args.noise, args.batch = 2048, 16384, "f32", "isotropic", 1
    if args.cpu_seconds is None:
        args.cpu_seconds = 10.0 if (args.D, args.N) == (128, 4096) else 0.0

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cpu_leg = None
    if args.cpu_seconds > 0 and world == 1 and rank == 0:
        cpu_leg = cpu_baseline(args.D, args.N, args.cpu_seconds, 123456)  # before any GPU initialisation
    import torch

    dist = None
    # BLR_BENCH_BACKEND=gloo + BLR_BENCH_SAME_DEVICE=1: validation of the N > 1 code path on a ONE-GPU box (both ranks on
    # cuda:0, the tiny collectives staged through the host).  The real runs use nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("BLR_BENCH_BACKEND", "nccl")
    if os.environ.get("BLR_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    import blr_amd
    from blr_amd import _abi

    h = _abi.Handle(local_rank)  # raises if the HIP extension or the GPU is missing: no fallback
    stream = torch.cuda.current_stream(dev)
    h.set_stream(stream.cuda_stream)  # 0 = the HIP null stream = torch's default stream
    h.set_async(True)

    B, D, N = args.batch, args.D, args.N
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    t_dt = torch.float64 if args.dtype == "f64" else torch.float32
    w_bytes = 8 if args.dtype == "f64" else 4
    diag = args.noise == "diagonal"

    # synthetic ColVecs design matrices: X ~ N(0,1), y = X'w* + sqrt(s) eps, prior mw = 0, Lw = I
    g = torch.Generator(device=dev).manual_seed(123456 + 1 + rank)
    X = torch.randn((B, N, D), generator=g, dtype=t_dt, device=dev)  # [N, D] row-major == D x N column-major
    wstar = torch.randn((B, D), generator=g, dtype=t_dt, device=dev)
    if diag:
        s = torch.exp(torch.randn((B, N), generator=g, dtype=t_dt, device=dev))
        sd = torch.sqrt(s)
    else:
        s = torch.full((1,), 0.1, dtype=t_dt, device=dev)
        sd = torch.sqrt(s)
    y = torch.einsum("bnd,bd->bn", X, wstar) + sd * torch.randn((B, N), generator=g, dtype=t_dt, device=dev)
    mw = torch.zeros((B, D), dtype=t_dt, device=dev)
    dprior = torch.ones((D,), dtype=t_dt, device=dev)
    mw_post = torch.empty((B, D), dtype=t_dt, device=dev)
    T_post = torch.empty((B, D, D), dtype=t_dt, device=dev)
    lp = torch.empty((B,), dtype=torch.float64, device=dev)
    info = torch.empty((B,), dtype=torch.int32, device=dev)
    lp_sum = torch.zeros((1,), dtype=torch.float64, device=dev)
    lp_all = torch.empty((B * world,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)

    noise_kind = _abi.NOISE_DIAGONAL if diag else _abi.NOISE_ISOTROPIC

    def fused_launch():
        h.posterior_batched(np_dt, _abi.MEM_DEVICE, _abi.LAYOUT_COLVECS, B, D, N, X.data_ptr(), D, N * D, y.data_ptr(), N,
                            noise_kind, s.data_ptr(), N if diag else 0, _abi.PRIOR_DIAGONAL, mw.data_ptr(), D,
                            dprior.data_ptr(), 1, 0, mw_post.data_ptr(), D, T_post.data_ptr(), D, D * D, None, D, D * D,
                            lp.data_ptr(), info.data_ptr())

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        fused_launch()
        if ev is not None:
            ev[1].record(stream)
        if dist is not None:
            # the path's only exchange: all-gather of the per-regressor log evidences (8 B each), then the SAME
            # fixed-order device sum on every rank -> identical bits for every rank count
            if backend == "nccl":
                dist.all_gather_into_tensor(lp_all, lp)
            else:  # host-staged collective (validation only)
                host = torch.empty(B * world, dtype=torch.float64)
                dist.all_gather_into_tensor(host, lp.cpu())
                lp_all.copy_(host)
            h.logpdf_sum(_abi.MEM_DEVICE, B * world, lp_all.data_ptr(), lp_sum.data_ptr())
        else:
            h.logpdf_sum(_abi.MEM_DEVICE, B, lp.data_ptr(), lp_sum.data_ptr())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    elapsed = float(t_el.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))

    # sanity: the timed work produced valid results
    assert int(info.abs().sum().item()) == 0, "a regressor failed to factorise"
    total_evidence = float(lp_sum.item())
    assert np.isfinite(total_evidence)

    if rank == 0:
        value = B * world * args.steps / elapsed
        fl = algorithmic_flops(D, N) * B
        by = algorithmic_bytes(D, N, w_bytes, diag) * B
        tf = fl / (kern_ms * 1e-3) / 1e12
        gbs = by / (kern_ms * 1e-3) / 1e9
        t_hbm = by / (PEAK_HBM_GBS * 1e9)
        t_mfma = fl / (PEAK_TF[args.dtype] * 1e12)
        if t_mfma >= t_hbm:
            roof = {"bound": "mfma", "achieved": tf, "peak": PEAK_TF[args.dtype], "unit": "TFLOP/s",
                    "frac": tf / PEAK_TF[args.dtype]}
        else:
            roof = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS}
        per_update = PMC_TRAFFIC_BYTES_PER_UPDATE.get((args.dtype, D, N, args.noise))
        roof.update({
            "traffic": per_update * B if per_update else None,
            "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, profiles/r01_pmc_summary.json"
                              if per_update else None,
            "algorithmic_bytes": by, "algorithmic_flops": fl,
            "mfma_peak_measured_TFLOPps": MEASURED_MFMA_TF[args.dtype],
            "mfma_frac_of_measured_peak": tf / MEASURED_MFMA_TF[args.dtype],
            "kernel": "fused_small_kernel",
            "kernel_ms_avg": kern_ms,
            "units_per_launch": B,
            "hbm_GBps": gbs, "hbm_frac": gbs / PEAK_HBM_GBS,
            "mfma_TFLOPps": tf, "mfma_frac": tf / PEAK_TF[args.dtype],
        })
        out = {
            "metric": "posterior-updates/sec + logpdf/sec at (D,N)",
            "value": value,
            "unit": "posterior-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{_config_name(D, N, args.dtype)}: independent regressors D={D}, N={N}, ColVecs, {args.noise} noise, {args.dtype}, "
                            f"Lw=I, fused posterior+logpdf",
                "D": D, "N": N, "batch_per_gpu": B, "global_batch": B * world,
                "sharding": f"regressors x{world}, no data-path collective; one all-gather of {B * world} doubles",
            },
            "roofline": roof,
            "total_log_evidence": total_evidence,
        }
        if cpu_leg is not None:
            out["cpu_baseline"] = cpu_leg
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    h.close()


if __name__ == "__main__":
    main()
