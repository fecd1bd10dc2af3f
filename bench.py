#!/usr/bin/env python3
"""bench.py -- posterior-updates/s of the fused MI355X posterior+logpdf path.

Headline workload (BASELINE.json configs[1], "c2"): independent regressors at D=128, N=4096, ColVecs, isotropic
noise, fp64, Lw = I, inputs resident in HBM.  One *step* = one pass of the hot path over one batch of
`--batch` regressors per GPU: ONE launch of the fused kernel (Gram + Cholesky + solves + evidence,
producing mw', T and logpdf for every regressor), the fixed-order device sum of the batch's log evidences,
and -- for N > 1 ranks -- the single all-gather of the per-rank log evidences (the only collective).
One "posterior update" = one regressor's full (mw', T, logpdf).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          (no launcher: bench.py starts that launcher itself as a child process and relays rank 0's
                                           line; a launcher that cannot start, or WORLD_SIZE != --gpus, is a non-zero exit)

Output on rank 0 (contract in the task statement): the LAST stdout line is ONE compact JSON object (< 3 KB: metric, value, unit,
n_gpus, steps, warmup, ms_per_step, dtype, config, roofline, cpu_baseline, scaling, vs_baseline).  At one GPU the other BASELINE
shapes and hot-path rows (c2 in fp32, c4, c3, c5 end to end, marginals, rand, gradient, ...) are timed the same way in the same
process AFTER the timed region; each prints one short JSON line ({"secondary": name, ms, per_s, bound, frac, traffic_x, kernel})
BEFORE the headline line, and their full records go to gpurun_out/bench_secondary_latest.json.
"""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# spec-sheet peaks (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters): HBM3E 8.0 TB/s; fp64 matrix 78.6 TF;
# fp32 matrix 157.3 TF
PEAK_HBM_GBS = 8000.0
PEAK_TF = {"f64": 78.6, "f32": 157.3}
PEAK_F16_TF = 2500.0  # dense fp16 / bf16 matrix peak (same guide; never the 2:1-sparsity figure)


def measured_matrix_peak(dtype):
    """What the matrix pipe sustains on this pool, from the committed microbenchmark logs (profiles/*_microbench_*.txt,
    written by tools/run_microbench.sh): f64 = tools/mfma_f64_probe.hip, all CUs, 2 waves per SIMD, accumulators in VGPRs;
    f32 = tools/mfma_peak.hip.  None when no log is present."""
    try:
        if dtype == "f64":
            f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_microbench_mfma_f64_probe.txt")))[-1]
            sect = open(f).read().split("---- all CUs, 2 waves/SIMD")[1]
            return float(re.search(r"acc VGPR\s.*?([\d.]+) TFLOP/s", sect).group(1)), os.path.relpath(f, ROOT)
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_microbench_mfma_peak.txt")))[-1]
        return float(re.search(r"f32 mfma 16x16x4, 2 wave/SIMD.*?([\d.]+) TFLOP/s", open(f).read()).group(1)), os.path.relpath(f, ROOT)
    except Exception:
        return None, None


def pmc_traffic_per_update(key):
    """HBM bytes per update from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE x2 on gfx950, + WRITE_SIZE, separate
    passes: tools/collect_profiles.sh -> tools/summarise_profiles.py).  None when the round's summary lacks the entry."""
    try:
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))[-1]
        d = json.load(open(f))
        e = d.get(key)
        if not e:
            return None, None
        # (a committed constant goes stale silently -- the summary records the kernel sources it was collected
        #  from; when the file that holds this entry's kernel has changed since, the source label says so)
        src = os.path.relpath(f, ROOT)
        if d.get("_collected_at_commit"):
            src += " @ " + d["_collected_at_commit"]
        shas = d.get("_kernel_source_sha16") or {}
        kern_file = {"fused_i8": "blr_fused_i8.hpp", "fused_small": "blr_fused_small.hpp", "fused_wave": "blr_fused_wave.hpp",
                     "gram_planes": "blr_planes.hpp"}
        for frag, fname in kern_file.items():
            if frag in key and fname in shas:
                import hashlib
                cur = hashlib.sha256(open(os.path.join(ROOT, "bayesianlinearregressors.jl_amd", "csrc", fname), "rb").read()).hexdigest()[:16]
                if cur != shas[fname]:
                    src += f" (STALE: {fname} has changed since this collection)"
        return float(e["hbm_bytes_per_launch"]) / float(e["units_per_launch"]), src
    except Exception:
        return None, None


def algorithmic_bytes(D, N, w, diag_noise, Din=None, dense_prior=False, lw_post=False):
    """Bytes one update of the TIMED call has to move (SURVEY.md 8(d) counted against what the call actually passes):
    reads X (D N; c5: the raw inputs D_in N + Omega D_in D + phases D), y (N), the noise (N or 1), the prior mean (D) and
    the prior precision (D for a diagonal prior, D^2 dense); writes mw' (D), T (D^2), Lw' (D^2, only when requested),
    logpdf (8) and info (4).  SURVEY's 2D^2 assumes a dense prior in AND Lw' out; the bench passes a diagonal prior and no
    Lw' buffer, so charging them would overstate the achieved bandwidth."""
    ns = N if diag_noise else 1
    x = (Din * N + Din * D + D) if Din is not None else D * N
    prior = D * D if dense_prior else D
    out = D + D * D + (D * D if lw_post else 0)
    return w * (x + N + ns + D + prior + out) + 12


def algorithmic_flops(D, N, Din=None):
    """D(D+1)N (symmetric-half SYRK) + 4DN + D^3/3 + 3D^2 + 5N per update (c5: + 2 D_in D N)."""
    f = D * (D + 1) * N + 4 * D * N + D**3 / 3 + 3 * D * D + 5 * N
    return f + (2 * Din * D * N if Din is not None else 0)


def physical_cores():
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        cps = int(re.search(r"Core\(s\) per socket:\s+(\d+)", out).group(1))
        sk = int(re.search(r"Socket\(s\):\s+(\d+)", out).group(1))
        return max(1, cps * sk)
    except Exception:
        return max(1, (os.cpu_count() or 2) // 2)


def _cpu_leg(D, N, seconds, seed, form, workers, threads):
    script = os.path.join(ROOT, "oracle", "cpu_baseline_worker.py")
    procs = [subprocess.Popen([sys.executable, script, str(D), str(N), str(seconds), str(seed + i), form, str(threads)],
                              stdout=subprocess.PIPE, text=True) for i in range(workers)]
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=seconds * 20 + 300)
        if p.returncode == 0 and out.strip():
            d, t = out.split()
            res.append((int(d), float(t)))
    return res


def cpu_baseline(D, N, seconds, seed):
    """The reference's algorithm restated (oracle/, NumPy/SciPy on OpenBLAS -- Julia is not installed), timed on the host:
      value                = literal op sequence (logpdf + posterior, reference :55-89), one process per PHYSICAL core x 1 BLAS thread
      direct_gram_value    = the one-pass Gram form (what a tuned CPU code would run), same process layout
      threaded_blas_value  = literal sequence, ONE process with OpenBLAS on all cores (how a single Julia session runs it)
    Child processes are started BEFORE this process touches the GPU; about `seconds` of wall time in total."""
    cores = physical_cores()
    per = max(1.0, seconds / 3.0)
    t0 = time.perf_counter()
    lit = _cpu_leg(D, N, per, seed, "literal", cores, 1)
    dirr = _cpu_leg(D, N, per, seed, "direct", cores, 1)
    thr = _cpu_leg(D, N, per, seed, "literal", 1, cores)
    wall = time.perf_counter() - t0
    if not lit:
        return None
    rate = lambda res: sum(d / t for d, t in res) if res else None
    return {
        "value": rate(lit),
        "unit": "posterior-updates/s",
        "cores": len(lit),
        "kind": "port",
        "direct_gram_value": rate(dirr),
        "threaded_blas_value": rate(thr),
        "sample": f"{sum(d for d, _ in lit)} regressors D={D} N={N} fp64, reference op sequence (logpdf+posterior) restated on "
                  f"NumPy/OpenBLAS: {len(lit)} procs x 1 BLAS thread, {per:.0f} s; direct_gram_value = one-pass Gram form, "
                  f"threaded_blas_value = 1 proc x {cores} BLAS threads; {wall:.0f} s wall",
    }


class Workload:
    """One BASELINE shape resident on the device + the call that runs one step of it."""

    def __init__(self, torch, _abi, h, dev, name, B, D, N, dtype, noise, seed, Din=None, mw_random=False, logpdf_only=False, block=None, rowvecs=False,
                 factor_prior=False, dense_prior=False, features="gauss"):
        self.name, self.B, self.D, self.N, self.dtype, self.noise, self.Din = name, B, D, N, dtype, noise, Din
        self.logpdf_only = logpdf_only
        self.rowvecs, self.factor_prior, self.dense_prior = rowvecs, factor_prior, dense_prior
        self.mw_random = mw_random
        self.torch, self._abi, self.h = torch, _abi, h
        t_dt = torch.float64 if dtype == "f64" else torch.float32
        self.np_dt = np.float64 if dtype == "f64" else np.float32
        self.w_bytes = 8 if dtype == "f64" else 4
        diag = noise == "diagonal"
        # synthetic ColVecs design matrices: X ~ N(0,1), y = X'w* + sqrt(s) eps, prior mw = 0, Lw = I
        def gen(nb, g):
            if Din is None:
                X = torch.randn((nb, N, D), generator=g, dtype=t_dt, device=dev)  # [N, D] row-major == D x N column-major
                if features == "student_t3":  # heavy-tailed features (Student-t, 3 degrees of freedom): rows outgrow bounds taken from their first columns
                    for c0 in range(0, nb, 256):
                        chi = torch.zeros_like(X[c0:c0 + 256])
                        for _ in range(3):
                            chi += torch.randn(chi.shape, generator=g, dtype=t_dt, device=dev) ** 2
                        X[c0:c0 + 256] /= torch.sqrt(chi / 3.0)
                        del chi
                wstar = torch.randn((nb, D), generator=g, dtype=t_dt, device=dev)
                mean = torch.einsum("bnd,bd->bn", X, wstar) if nb * N * D < (1 << 32) else torch.stack([X[b] @ wstar[b] for b in range(nb)])
            else:  # c5: raw inputs D_in x N, random-Fourier basis on the device
                X = torch.randn((nb, N, Din), generator=g, dtype=t_dt, device=dev)
                mean = torch.zeros((nb, N), dtype=t_dt, device=dev)
            if diag:
                sv = torch.exp(torch.randn((nb, N), generator=g, dtype=t_dt, device=dev))
                sd = torch.sqrt(sv)
            else:
                sv, sd = None, float(np.sqrt(0.1))
            return X, sv, mean + sd * torch.randn((nb, N), generator=g, dtype=t_dt, device=dev)

        g = torch.Generator(device=dev).manual_seed(seed)
        if Din is not None:
            self.Omega = torch.randn((D, Din), generator=g, dtype=t_dt, device=dev)  # [D, Din] row-major == Din x D column-major
            self.phase = 2 * np.pi * torch.rand((D,), generator=g, dtype=t_dt, device=dev)
        if block is None:
            self.X, sv, self.y = gen(B, g)
        else:
            # strong scaling: this rank's contiguous block [lo, hi) of a FIXED global batch.  Regressor i is the same on every
            # rank count: chunk c of `ch` regressors is drawn from a generator seeded with seed + c, whoever holds it.
            lo, hi, ch = block
            assert hi - lo == B
            self.X = torch.empty((B, N, D if Din is None else Din), dtype=t_dt, device=dev)
            self.y = torch.empty((B, N), dtype=t_dt, device=dev)
            sv = torch.empty((B, N), dtype=t_dt, device=dev) if diag else None
            for c in range(lo // ch, (hi + ch - 1) // ch):
                Xc, sc, yc = gen(ch, torch.Generator(device=dev).manual_seed(seed + 1000 + c))
                a0, a1 = max(lo, c * ch), min(hi, (c + 1) * ch)
                self.X[a0 - lo:a1 - lo] = Xc[a0 - c * ch:a1 - c * ch]
                self.y[a0 - lo:a1 - lo] = yc[a0 - c * ch:a1 - c * ch]
                if diag:
                    sv[a0 - lo:a1 - lo] = sc[a0 - c * ch:a1 - c * ch]
                del Xc, sc, yc
        self.s = sv if diag else torch.full((1,), 0.1, dtype=t_dt, device=dev)
        # prior mean: 0 (SURVEY.md 8(d) first variant: the zero-mean fast paths) or ~ N(0, I) (second variant; the reference's toy
        # problems draw mw = randn(D), test/test_utils.jl:6)
        self.mw = torch.randn((B, D), generator=g, dtype=t_dt, device=dev) if mw_random else torch.zeros((B, D), dtype=t_dt, device=dev)
        self.dprior = torch.ones((D,), dtype=t_dt, device=dev)
        if rowvecs:  # the same design matrices stored N x D column-major (RowVecs: a feature's observations contiguous)
            self.X = self.X.transpose(1, 2).contiguous()
        if factor_prior:  # a prior given by its upper factor U (Lw = U'U, a PDMat / a carried-forward posterior), shared by the batch;
            # [D, D] row-major lower triangle == column-major upper U
            self.Uprior = torch.tril(torch.randn((D, D), generator=g, dtype=t_dt, device=dev) / float(np.sqrt(D)), -1) + 1.5 * torch.eye(D, dtype=t_dt, device=dev)
        if dense_prior:  # one dense symmetric precision PER regressor, Lw = B B' / D + I (the reference's toy priors, test/test_utils.jl:6-8)
            Bm = torch.randn((B, D, D), generator=g, dtype=t_dt, device=dev) / float(np.sqrt(D))
            self.Lprior = torch.bmm(Bm, Bm.transpose(1, 2)) + torch.eye(D, dtype=t_dt, device=dev)
            del Bm
        self.mw_post = torch.empty((B, D), dtype=t_dt, device=dev)
        self.T_post = torch.empty((B, D, D), dtype=t_dt, device=dev)
        self.lp = torch.empty((B,), dtype=torch.float64, device=dev)
        self.info = torch.empty((B,), dtype=torch.int32, device=dev)
        self.noise_kind = _abi.NOISE_DIAGONAL if diag else _abi.NOISE_ISOTROPIC
        self.diag = diag

    def launch(self):
        a, B, D, N = self._abi, self.B, self.D, self.N
        if self.Din is None:
            pk, pr, ldl = (a.PRIOR_UPPER_FACTOR, self.Uprior, D) if self.factor_prior else (a.PRIOR_DIAGONAL, self.dprior, 1)
            sLw = 0
            if self.dense_prior:
                pk, pr, ldl, sLw = a.PRIOR_DENSE, self.Lprior, D, D * D
            self.h.posterior_batched(self.np_dt, a.MEM_DEVICE, a.LAYOUT_ROWVECS if self.rowvecs else a.LAYOUT_COLVECS, B, D, N, self.X.data_ptr(),
                                     N if self.rowvecs else D, N * D, self.y.data_ptr(), N,
                                     self.noise_kind, self.s.data_ptr(), N if self.diag else 0, pk, self.mw.data_ptr(), D,
                                     pr.data_ptr(), ldl, sLw, None if self.logpdf_only else self.mw_post.data_ptr(), D,
                                     None if self.logpdf_only else self.T_post.data_ptr(), D, D * D, None, D,
                                     D * D, self.lp.data_ptr(), self.info.data_ptr())
        else:
            self.h.posterior_rff(self.np_dt, a.MEM_DEVICE, self.Din, D, N, self.X.data_ptr(), self.Din, self.Omega.data_ptr(), self.Din,
                                 self.phase.data_ptr(), float(np.sqrt(2.0 / D)), self.y.data_ptr(), self.noise_kind, self.s.data_ptr(),
                                 a.PRIOR_DIAGONAL, self.mw.data_ptr(), self.dprior.data_ptr(), 1, self.mw_post.data_ptr(),
                                 self.T_post.data_ptr(), D, None, D, self.lp.data_ptr(), self.info.data_ptr())

    def kernel_name(self):
        """The kernel family the dispatcher took for the most recent launch of this workload (blr_last_route): asked, not re-derived."""
        return self.h.last_route()

    def roofline(self, ms):
        fl = algorithmic_flops(self.D, self.N, self.Din) * self.B
        by = algorithmic_bytes(self.D, self.N, self.w_bytes, self.diag, self.Din, dense_prior=self.dense_prior) * self.B
        if self.logpdf_only:  # no mw', no T written
            by -= self.w_bytes * (self.D + self.D * self.D) * self.B
        kern = self.kernel_name()
        r = roofline_of(fl, by, self.dtype, ms, kern, i8_cols=self.N * self.B if kern == "fused_i8_kernel" else None, i8_diag=self.diag)
        r.update({"kernel_ms_avg": ms, "units_per_launch": self.B})
        return r


I8_OPS_PER_COLUMN = 174 * 2048.0  # int8 route: MFMAs (32 x 32 x 32: 2 x 32 x 32 operations per column each) per k-step; fallback if the header cannot be read
PEAK_I8_TOPS = 5000.0                 # dense int8 peak (2 x the bf16 peak per clock; MI355X_MICROARCH.md matrix-core table)


def i8_ops_per_column(diag=False):
    """int8 operations per column of the int8 route AS BUILT: csrc/blr_fused_i8.hpp states the MFMAs (32 x 32 x 32) per 32-column
    k-step of its two plans in kI8MfmaPerKstep (isotropic noise) and kI8MfmaPerKstepDiag (diagonal noise)."""
    try:
        src = open(os.path.join(ROOT, "bayesianlinearregressors.jl_amd", "csrc", "blr_fused_i8.hpp")).read()
        m = re.search(r"constexpr int kI8MfmaPerKstep%s\s*=\s*(\d+)" % ("Diag" if diag else ""), src)
        if m:
            return int(m.group(1)) * 2048.0
    except Exception:
        pass
    return I8_OPS_PER_COLUMN


def roofline_of(flops, nbytes, dtype, ms, kernel, i8_cols=None, i8_diag=False):
    """bound = whichever of t_HBM (8 TB/s) and t_matrix is longer for the ALGORITHMIC bytes / flops of one call (stated per entry in
    DESIGN.md 5); frac = that time / measured time.  t_matrix is flops / the dense matrix peak of the dtype -- except on the int8
    route (i8_cols = columns streamed per call), whose Gram runs on the int8 cores: there it is int8 operations / 5 POP/s, which is
    shorter than streaming X, so the route is HBM-bound; every entry that takes it gets the same three fractions (hbm_frac,
    int8_frac, f64_equiv_frac = the fp64-matrix-pipe rate the same work would need) so rounds and routes stay comparable."""
    sec = ms * 1e-3
    tf = flops / sec / 1e12
    gbs = nbytes / sec / 1e9
    t_hbm = nbytes / (PEAK_HBM_GBS * 1e9)
    t_mat = flops / (PEAK_TF[dtype] * 1e12)
    extra = {}
    if i8_cols is not None:
        ops = i8_ops_per_column(i8_diag) * i8_cols
        t_mat = ops / (PEAK_I8_TOPS * 1e12)
        extra = {"int8_TOPps": ops / sec / 1e12, "int8_frac": ops / sec / (PEAK_I8_TOPS * 1e12), "f64_equiv_frac": tf / PEAK_TF[dtype]}
    planes = {"gram_planes4_kernel": 3.0, "gram_planes_kernel<2>": 3.0, "gram_planes_kernel<3>": 6.0}.get(kernel)
    if planes is not None and dtype == "f32":
        # the fp32 Gram from pre-split 16-bit planes (csrc/blr_planes.hpp): `planes` products of the half-precision matrix instruction
        # (2.5 PFLOP/s dense) per fp32 product -- priced against the pipe it runs on; f32_equiv_frac keeps the round-5 convention
        t_mat = planes * flops / (PEAK_F16_TF * 1e12)
        extra = {"f16_TFLOPps": planes * tf, "f16_frac": planes * tf / PEAK_F16_TF, "f32_equiv_frac": tf / PEAK_TF[dtype]}
    if t_mat >= t_hbm:
        if planes is not None and dtype == "f32":
            r = {"bound": "mfma", "achieved": extra["f16_TFLOPps"], "peak": PEAK_F16_TF, "unit": "TFLOP/s (fp16 / bf16 products)", "frac": extra["f16_frac"]}
        elif i8_cols is not None:
            r = {"bound": "mfma", "achieved": extra["int8_TOPps"], "peak": PEAK_I8_TOPS, "unit": "TOP/s (int8)", "frac": extra["int8_frac"]}
        else:
            r = {"bound": "mfma", "achieved": tf, "peak": PEAK_TF[dtype], "unit": "TFLOP/s", "frac": tf / PEAK_TF[dtype]}
    else:
        r = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS}
    r.update({"kernel": kernel, "hbm_frac": gbs / PEAK_HBM_GBS, "mfma_frac": tf / PEAK_TF[dtype], "algorithmic_bytes": nbytes,
              "algorithmic_flops": flops})
    r.update(extra)
    return r


def _sig(x, n=4):
    return float(f"{x:.{n}g}") if isinstance(x, float) else x


def secondary_line(name, e):
    """One short JSON line (<= 256 B) per secondary entry, printed BEFORE the headline: a tail of stdout still reads them."""
    if "error" in e:
        return json.dumps({"secondary": name, "error": e["error"][:120]})
    r = e["roofline"]
    tx = (r["traffic"] / r["algorithmic_bytes"]) if r.get("traffic") else None
    d = {"secondary": name, "ms": _sig(e["ms"]), "per_s": _sig(e["per_s"]), "unit": e["unit"], "bound": r["bound"], "frac": _sig(r["frac"], 3),
         "traffic_x": _sig(tx, 3) if tx else None, "kernel": r["kernel"][:72]}
    if "int8_frac" in r:
        d["int8_frac"], d["f64_equiv_frac"] = _sig(r["int8_frac"], 3), _sig(r["f64_equiv_frac"], 3)
    if "f32_equiv_frac" in r:
        d["f32_equiv_frac"] = _sig(r["f32_equiv_frac"], 3)
    return json.dumps(d)


HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config", "roofline", "cpu_baseline", "total_log_evidence", "preheat_s", "sustained", "secondary_file", "n_secondary")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg", "kernel_ms_min", "kernel_ms_median",
                 "units_per_launch", "algorithmic_bytes", "algorithmic_flops", "hbm_frac", "mfma_frac", "int8_frac", "f64_equiv_frac",
                 "traffic_source", "matrix_peak_measured_TFLOPps", "mfma_frac_of_measured_peak")


def headline_line(out):
    """The driver's line: the contract's keys only, floats at 6 significant digits, < 3 KB whatever the run added to `out`."""
    d = {k: out[k] for k in HEADLINE_KEYS if k in out}
    if "roofline" in d:
        d["roofline"] = {k: (_sig(v, 6) if isinstance(v, float) else v) for k, v in out["roofline"].items() if k in ROOFLINE_KEYS}
    if "cpu_baseline" in d:
        d["cpu_baseline"] = {k: (_sig(v, 6) if isinstance(v, float) else v) for k, v in out["cpu_baseline"].items()}
        d["cpu_baseline"]["sample"] = d["cpu_baseline"].get("sample", "")[:400]
    for k in ("value", "ms_per_step", "total_log_evidence"):
        if isinstance(d.get(k), float):
            d[k] = _sig(d[k], 9)
    line = json.dumps(d)
    assert len(line) < 3072, f"headline line is {len(line)} bytes"
    return line


class Op:
    """One secondary hot-path operation: `fn` enqueues ONE call on the launch stream; units / flops / nbytes are per call."""

    def __init__(self, workload, fn, units, unit, flops, nbytes, dtype, kernel, check, steps=20, keep=()):
        self.workload, self.fn, self.units, self.unit, self.flops, self.nbytes = workload, fn, units, unit, flops, nbytes
        self.dtype, self.kernel, self.check, self.steps, self.keep = dtype, kernel, check, steps, keep


def secondary_ops(torch, _abi, h, dev):
    """name -> builder of an Op.  Every hot-path row of SURVEY.md 8(a) other than `posterior` (a7 mean, a8 var, a9 rand, a4 logpdf
    alone) and the 8(f) rows (gradient, shared-X evidence, rank-k update of a resident state), on device-resident synthetic inputs."""
    a = _abi

    def tdt(dt):
        return (torch.float64, np.float64, 8) if dt == "f64" else (torch.float32, np.float32, 4)

    def factor(g, B, D, t):  # well-conditioned upper factors U (column-major storage), as tools/marginals_bench.py
        U = torch.triu(torch.randn((B, D, D), generator=g, dtype=t, device=dev)) / D**0.5 + 2 * torch.eye(D, dtype=t, device=dev)
        return U, U.transpose(1, 2).contiguous()

    def marginals(B, D, N, dt, mean_only):
        t, nd, w = tdt(dt)
        g = torch.Generator(device=dev).manual_seed(11)
        X = torch.randn((B, N, D), generator=g, dtype=t, device=dev)
        mw = torch.randn((B, D), generator=g, dtype=t, device=dev)
        U, Ucm = factor(g, B, D, t)
        s = torch.full((1,), 0.1, dtype=t, device=dev)
        mean = torch.empty((B, N), dtype=t, device=dev)
        var = torch.empty((B, N), dtype=t, device=dev)
        info = torch.zeros(B, dtype=torch.int32, device=dev)

        def fn():
            h.marginals_batched(nd, a.MEM_DEVICE, a.LAYOUT_COLVECS, B, D, N, X.data_ptr(), D, N * D, a.NOISE_ISOTROPIC, s.data_ptr(), 0,
                                a.PRIOR_UPPER_FACTOR, mw.data_ptr(), D, Ucm.data_ptr(), D, D * D, mean.data_ptr(), N,
                                None if mean_only else var.data_ptr(), N, info.data_ptr())

        def check():
            m_ref = X[0].double() @ mw[0].double()
            assert float((mean[0].double() - m_ref).abs().max() / m_ref.abs().max()) < (1e-12 if dt == "f64" else 1e-4)
            if not mean_only:
                al = torch.linalg.solve_triangular(U[0].double().T, X[0].T.double(), upper=False)
                v_ref = (al * al).sum(0) + 0.1
                assert float(((var[0].double() - v_ref).abs() / v_ref).max()) < (1e-11 if dt == "f64" else 2e-4)

        # mean: 2DN flops on DN elements (a7, pure bandwidth); var: the D^2 N triangular solve (a8: D^2 flops per input) + 2DN
        flops = B * N * (2 * D if mean_only else D * D + 4 * D)
        nbytes = w * B * (N * D + D + (0 if mean_only else D * D) + N * (1 if mean_only else 2))
        kern = ("mean_stream_kernel" if mean_only else "marg_blocksub_kernel") if D > 128 else ("marginals_mfma_kernel" if mean_only else "marginals_gemm_kernel")
        return Op(f"B={B}, D={D}, N={N}, {dt}: {'mean' if mean_only else 'mean + var'} of the marginals, factor prior (PDMat / posterior)",
                  fn, B * N, "marginals/s", flops, nbytes, dt, kern, check, keep=(X, mw, Ucm, s, mean, var, info))

    def rand(D, N, S, dt):
        t, nd, w = tdt(dt)
        g = torch.Generator(device=dev).manual_seed(12)
        X = torch.randn((N, D), generator=g, dtype=t, device=dev)
        Z1 = torch.randn((S, D), generator=g, dtype=t, device=dev)
        Z2 = torch.randn((S, N), generator=g, dtype=t, device=dev)
        Y = torch.empty((S, N), dtype=t, device=dev)
        s = torch.full((1,), 0.1, dtype=t, device=dev)
        mw = torch.randn((D,), generator=g, dtype=t, device=dev)
        U, Ucm = factor(g, 1, D, t)

        def fn():
            h.rand(nd, a.MEM_DEVICE, a.LAYOUT_COLVECS, D, N, S, X.data_ptr(), D, a.NOISE_ISOTROPIC, s.data_ptr(), a.PRIOR_UPPER_FACTOR,
                   mw.data_ptr(), Ucm.data_ptr(), D, Z1.data_ptr(), D, Z2.data_ptr(), N, Y.data_ptr(), N)

        def check():
            W = mw.double()[:, None] + torch.linalg.solve_triangular(U[0].double(), Z1.double().T, upper=True)
            ref = X.double() @ W + (0.1 ** 0.5) * Z2.double().T
            assert float((Y.double().T - ref).abs().max() / ref.abs().max()) < (1e-11 if dt == "f64" else 2e-4)

        return Op(f"D={D}, N={N}, {S} draws, {dt}: rand(rng, fx, S) with the normals on the device (weight solve + projection + noise)", fn,
                  N * S, "outputs/s", 2.0 * D * N * S + D * D * S, w * (D * N + 2 * N * S + 2 * D * S + D * D + D), dt,
                  "rand_project_mfma_kernel", check, keep=(X, Z1, Z2, Y, s, mw, Ucm))

    def grad(B, D, N, dt, option=None):
        t, nd, w = tdt(dt)
        g = torch.Generator(device=dev).manual_seed(13)
        X = torch.randn((B, N, D), generator=g, dtype=t, device=dev)
        y = torch.randn((B, N), generator=g, dtype=t, device=dev)
        s = torch.full((1,), 0.1, dtype=t, device=dev)
        mw = torch.randn((B, D), generator=g, dtype=t, device=dev)
        d = torch.ones((D,), dtype=t, device=dev)
        lp = torch.zeros(B, dtype=torch.float64, device=dev)
        info = torch.zeros(B, dtype=torch.int32, device=dev)
        dX, dy, ds, dmw, mwp = torch.empty_like(X), torch.empty_like(y), torch.empty_like(y), torch.empty_like(mw), torch.empty_like(mw)
        Ai = torch.empty((B, D, D), dtype=t, device=dev)

        def fn():
            h.logpdf_grad_batched(nd, a.MEM_DEVICE, a.LAYOUT_COLVECS, B, D, N, X.data_ptr(), D, N * D, y.data_ptr(), N, a.NOISE_ISOTROPIC,
                                  s.data_ptr(), 0, a.PRIOR_DIAGONAL, mw.data_ptr(), D, d.data_ptr(), 1, 0, lp.data_ptr(), dX.data_ptr(), D,
                                  N * D, dy.data_ptr(), N, ds.data_ptr(), N, dmw.data_ptr(), D, mwp.data_ptr(), D, Ai.data_ptr(), D, D * D,
                                  info.data_ptr())

        if option:  # the same call with a run-time switch of the handle set for its duration (A/B entry)
            fn0 = fn

            def fn():
                h.set_option(option, "1")
                try:
                    fn0()
                finally:
                    h.set_option(option, None)

        def check():
            assert int(info.abs().sum().item()) == 0 and bool(torch.isfinite(lp).all().item()) and bool(torch.isfinite(dX).all().item())

        # value (SYRK half D(D+1)N + 4DN + chol D^3/3) + A^-1 (2D^3/3) + the dX pass (A^-1 X: 2 D^2 N) + O(DN) vector work
        flops = B * (D * (D + 1) * N + 2.0 * D * D * N + D**3 + 12.0 * D * N)
        nbytes = w * B * (2 * N * D + 4 * N + D * D + 4 * D)
        return Op(f"B={B}, D={D}, N={N}, {dt}: value + gradient of the log marginal likelihood w.r.t. X, y, s, mw (+ A^-1)"
                  + (f", handle option {option}" if option else ""), fn, B,
                  "evaluations/s", flops, nbytes, dt, "logpdf_grad_kernel" if option == "NO_GRAD_GEMM" else "grad_gemm_kernel", check, steps=10, keep=(X, y, s, mw, d, lp, info, dX, dy, ds, dmw, mwp, Ai))

    def multi(D, N, S, dt):
        t, nd, w = tdt(dt)
        g = torch.Generator(device=dev).manual_seed(14)
        X = torch.randn((N, D), generator=g, dtype=t, device=dev)
        Y = torch.randn((S, N), generator=g, dtype=t, device=dev)
        s = torch.full((1,), 0.1, dtype=t, device=dev)
        mw = torch.randn((D,), generator=g, dtype=t, device=dev)
        d = torch.ones((D,), dtype=t, device=dev)
        lp = torch.zeros(S, dtype=torch.float64, device=dev)
        info = torch.zeros(S, dtype=torch.int32, device=dev)

        def fn():
            h.logpdf_multi(nd, a.MEM_DEVICE, a.LAYOUT_COLVECS, D, N, S, X.data_ptr(), D, Y.data_ptr(), N, a.NOISE_ISOTROPIC, s.data_ptr(),
                           a.PRIOR_DIAGONAL, mw.data_ptr(), d.data_ptr(), 1, lp.data_ptr(), None, D, info.data_ptr())

        def check():
            assert int(info.abs().sum().item()) == 0 and bool(torch.isfinite(lp).all().item())

        flops = D * (D + 1) * N + 2.0 * D * N * S + D**3 / 3 + 2.0 * D * D * S + 4.0 * D * N
        return Op(f"D={D}, N={N}, S={S} columns of Y sharing X, {dt}: logpdf(fx, Y::Matrix)", fn, S, "evidences/s", flops,
                  w * (D * N + N * S + 2 * D + 1) + 8 * S, dt, "gram_tile_kernel", check, steps=10, keep=(X, Y, s, mw, d, lp, info))

    def update(B, D, k, dt):
        t, nd, w = tdt(dt)
        g = torch.Generator(device=dev).manual_seed(15)
        U = torch.triu(torch.randn((B, D, D), generator=g, dtype=t, device=dev)) * (0.3 / D**0.5)
        U = U + torch.diag_embed(1.0 + U.diagonal(dim1=1, dim2=2).abs())
        T0 = U.transpose(1, 2).contiguous()
        Tm = T0.clone()
        X = torch.randn((B, k, D), generator=g, dtype=t, device=dev)
        y = torch.randn((B, k), generator=g, dtype=t, device=dev)
        s = torch.full((1,), 0.5, dtype=t, device=dev)
        mw = torch.zeros((B, D), dtype=t, device=dev)
        lp = torch.zeros(B, dtype=torch.float64, device=dev)
        info = torch.zeros(B, dtype=torch.int32, device=dev)

        def fn():  # the state keeps absorbing the same k observations: precision grows, always positive definite
            h.update_factor(nd, a.MEM_DEVICE, a.LAYOUT_COLVECS, B, D, k, X.data_ptr(), D, k * D, y.data_ptr(), k, a.NOISE_ISOTROPIC, s.data_ptr(),
                            0, mw.data_ptr(), D, Tm.data_ptr(), D, D * D, lp.data_ptr(), info.data_ptr())

        def check():
            assert int(info.abs().sum().item()) == 0 and bool(torch.isfinite(lp).all().item()) and bool(torch.isfinite(Tm).all().item())

        # state read + written in place (upper triangle of T, mw) + the k new columns; the re-factorisation route costs D^3/3 + k D^2
        nbytes = w * B * (D * (D + 1) + 2 * D + k * D + 2 * k) + 12 * B
        flops = B * (D**3 / 3 + 2.0 * k * D * D + 3.0 * D * D)
        return Op(f"B={B}, D={D}, k={k}, {dt}: in-place rank-k update of a resident posterior (mw, T) + evidence increment", fn, B,
                  "updates/s", flops, nbytes, dt, "rank1_sweep_kernel" if k <= 1 else "fused_small_kernel (factor prior, in place)", check,
                  keep=(Tm, T0, X, y, s, mw, lp, info))

    def post(name, b, d, n, dt, noise, din=None, steps=20, option=None, **kw):
        def build():
            w2 = Workload(torch, a, h, dev, name, b, d, n, dt, noise, 123456 + 7, din, **kw)
            launch, kern = w2.launch, "posterior"
            if option:  # the same workload with a run-time switch of the handle set for the duration of each call (A/B entry)

                def launch():
                    h.set_option(option, "1")
                    try:
                        w2.launch()
                    finally:
                        h.set_option(option, None)

            def check():
                assert int(w2.info.abs().sum().item()) == 0 and bool(torch.isfinite(w2.lp).all().item())

            r = {"algorithmic_flops": algorithmic_flops(w2.D, w2.N, w2.Din) * w2.B,
                 "algorithmic_bytes": (algorithmic_bytes(w2.D, w2.N, w2.w_bytes, w2.diag, w2.Din, dense_prior=w2.dense_prior)
                                       - (w2.w_bytes * (w2.D + w2.D * w2.D) if w2.logpdf_only else 0)) * w2.B}
            tag = (", prior mean ~ N(0, I)" if kw.get("mw_random") else "") + (", logpdf only (no mw', no T)" if kw.get("logpdf_only") else "") \
                + (", RowVecs storage" if kw.get("rowvecs") else "") + (", prior by its upper factor" if kw.get("factor_prior") else "") \
                + (", dense prior precision per regressor" if kw.get("dense_prior") else "") \
                + (", Student-t(3) features" if kw.get("features") == "student_t3" else "") \
                + (f", handle option {option}" if option else "")
            return Op(f"B={b}, D={d}, N={n}, {dt}, {noise} noise" + (f", D_in={din} random-Fourier features" if din else "") + tag,
                      launch, b, "updates/s", r["algorithmic_flops"], r["algorithmic_bytes"], dt, kern, check, steps=steps, keep=(w2,))
        return build

    ops = {
        "c2_f32": post("c2_f32", 4096, 128, 4096, "f32", "isotropic"),
        # the driver line's workload on the fp64 matrix pipe (fused_small_kernel) instead of the int8-sliced Gram: same box, same data
        "c2_f64_fp64_kernel": post("c2", 4096, 128, 4096, "f64", "isotropic", option="NO_I8_GRAM"),
        "c2_f64_mw": post("c2_f64_mw", 4096, 128, 4096, "f64", "isotropic", mw_random=True),
        "c2_f64_diag_noise": post("c2_f64_diag", 4096, 128, 4096, "f64", "diagonal"),
        "c2_f64_factor_prior": post("c2_f64_factor", 4096, 128, 4096, "f64", "isotropic", factor_prior=True),
        "c2_f64_rowvecs": post("c2_f64_rowvecs", 4096, 128, 4096, "f64", "isotropic", rowvecs=True),
        "c2_f64_dense_prior": post("c2_f64_dense", 4096, 128, 4096, "f64", "isotropic", dense_prior=True),
        # what the DEFAULT route costs on heavy-tailed features (Student-t(3): the int8 route's row bounds do not hold; 8192 regressors: the
        # probe slice of 256 hands back more than a quarter and the rest of the batch goes to the fp64 kernel directly)
        "c2_f64_heavy_tail": post("c2_f64_heavy_tail", 8192, 128, 4096, "f64", "isotropic", steps=6, features="student_t3"),
        "c4_f64": post("c4_f64", 8192, 64, 1024, "f64", "isotropic"),
        "c4_f32": post("c4_f32", 8192, 64, 1024, "f32", "isotropic"),
        # the per-GPU blocks of config 4 (8192 regressors) on 2 / 4 / 8 GPUs: the expected strong-scaling curve
        "c4_f64_B4096": post("c4_f64_B4096", 4096, 64, 1024, "f64", "isotropic"),
        "c4_f64_B2048": post("c4_f64_B2048", 2048, 64, 1024, "f64", "isotropic"),
        "c4_f64_B1024": post("c4_f64_B1024", 1024, 64, 1024, "f64", "isotropic"),
        "c3_f32": post("c3_f32", 1, 1024, 65536, "f32", "diagonal"),
        "c3_f32_mw": post("c3_f32_mw", 1, 1024, 65536, "f32", "diagonal", mw_random=True),
        "logpdf_only_c3_f32": post("logpdf_only_c3", 1, 1024, 65536, "f32", "diagonal", logpdf_only=True),
        "c5_f32_end_to_end": post("c5_f32_end_to_end", 1, 2048, 16384, "f32", "isotropic", 8),
        # batches at D > 128: the regressors go through every launch of the update together (posterior_large_group)
        "c3_f32_B8": post("c3_f32_B8", 8, 1024, 65536, "f32", "diagonal", steps=10),
        "c5_shape_f32_B8": post("c5_shape_f32_B8", 8, 2048, 16384, "f32", "isotropic", steps=10),
        "marginals_mean_c2_f64": lambda: marginals(64, 128, 4096, "f64", True),
        "marginals_var_c2_f64": lambda: marginals(64, 128, 4096, "f64", False),
        "marginals_var_c2_f32": lambda: marginals(64, 128, 4096, "f32", False),
        "marginals_mean_c3_f32": lambda: marginals(1, 1024, 65536, "f32", True),
        "marginals_var_c3_f32": lambda: marginals(1, 1024, 65536, "f32", False),
        "marginals_var_D512_B16_f32": lambda: marginals(16, 512, 16384, "f32", False),  # a batch at D > 128: one set of launches
        "rand_c2_f64_S64": lambda: rand(128, 4096, 64, "f64"),
        "rand_c3_f32_S64": lambda: rand(1024, 65536, 64, "f32"),
        "logpdf_grad_c2_f64": lambda: grad(1024, 128, 4096, "f64"),
        # the same call on the sweep kernel (two blocked triangular sweeps per tile through LDS) instead of the product form
        "logpdf_grad_c2_f64_sweep_kernel": lambda: grad(1024, 128, 4096, "f64", option="NO_GRAD_GEMM"),
        "logpdf_grad_c2_f32": lambda: grad(1024, 128, 4096, "f32"),
        "logpdf_multi_c3_f32_S64": lambda: multi(1024, 65536, 64, "f32"),
        "update_factor_D128_k1_f64": lambda: update(2048, 128, 1, "f64"),
        "update_factor_D128_k16_f64": lambda: update(2048, 128, 16, "f64"),
    }
    return ops


def run_secondary(torch, _abi, h, dev, stream, only=None):
    sec = {}
    for name, build in secondary_ops(torch, _abi, h, dev).items():
        if only is not None and name not in only:
            continue
        try:
            op = build()
            torch.cuda.synchronize(dev)
            wall, ms = timed(torch, stream, dev, op.fn, op.steps, 3)
            op.check()
            kern = op.kernel
            if op.unit in ("updates/s", "evidences/s") and not op.kernel.startswith("rank1") and "in place" not in op.kernel:
                kern = h.last_route()  # posterior workloads (the multi-output evidence rides one): the route the dispatcher took on the last call
            i8_cols, i8_diag = None, False
            if kern == "fused_i8_kernel":
                w2 = op.keep[0]
                i8_cols, i8_diag = w2.N * w2.B, w2.diag
            r = roofline_of(op.flops, op.nbytes, op.dtype, ms, kern, i8_cols=i8_cols, i8_diag=i8_diag)
            # HBM bytes of one CALL by PMC (all of the call's kernels; tools/collect_profiles.sh runs `--secondary-only <name>` under
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE); summaries of earlier rounds hold the dominant kernel of four entries per update
            per_call, src = pmc_traffic_per_update(name + "_hbm")
            legacy = {"c2_f32": "c2_f32_fused_small_kernel_hbm", "c4_f64": "c4_fused_wave_kernel_hbm", "c3_f32": "c3_gram_planes_kernel_hbm",
                      "c5_f32_end_to_end": "c5_gram_planes_kernel_hbm"}
            if per_call is None and name in legacy:
                per_call, src = pmc_traffic_per_update(legacy[name])
                if per_call is not None and name in ("c2_f32", "c4_f64"):
                    per_call *= op.units
            r["traffic"], r["traffic_source"] = per_call, src
            print(secondary_line(name, {"ms": ms, "unit": op.unit, "per_s": op.units / (ms * 1e-3), "roofline": r}), flush=True)
            sec[name] = {"workload": op.workload, "ms": ms, "unit": op.unit, "per_s": op.units / (ms * 1e-3),
                         "wall_per_s": op.units * op.steps / wall, "calls": op.steps + 3, "roofline": r}
            if op.unit == "updates/s":
                sec[name]["updates_per_s"] = sec[name]["per_s"]
            del op
            torch.cuda.empty_cache()
        except Exception as e:  # a secondary entry must never take the headline line down
            sec[name] = {"error": f"{type(e).__name__}: {e}"}
            print(secondary_line(name, sec[name]), flush=True)
            torch.cuda.empty_cache()
    return sec


def timed(torch, stream, dev, fn, steps, warmup):
    """(wall seconds for `steps` calls, mean ms between HIP events recorded around each call ON THE LAUNCH STREAM)"""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0, float(np.mean([a.elapsed_time(b) for a, b in ev]))


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks the way the driver would (torch.distributed.run, one process
    per GPU, rendezvous on 127.0.0.1) as a CHILD process -- this process has not initialised the GPU and never does -- relay its
    output line by line (rank 0's headline stays the last stdout line) and return its exit code; a launcher that cannot start
    is a non-zero exit, not a one-rank run.  BLR_BENCH_LAUNCHER overrides the launcher module (tests force a spawn failure with it)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    launcher = os.environ.get("BLR_BENCH_LAUNCHER", "torch.distributed.run")
    cmd = [sys.executable, "-m", launcher, "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"bench.py: --gpus {n} without a launcher: spawning {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    except OSError as e:
        print(f"bench.py: could not start the launcher: {e}", file=sys.stderr, flush=True)
        return 1
    last = None
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        if line.strip():
            last = line
    rc = proc.wait()
    if rc == 0:
        try:
            ok = json.loads(last).get("n_gpus") == n
        except Exception:
            ok = False
        if not ok:
            print(f"bench.py: the spawned job did not end with a headline for n_gpus = {n}", file=sys.stderr, flush=True)
            return 1
    return rc


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
    ap.add_argument("--preheat-seconds", type=float, default=None,
                    help="untimed back-to-back steps BEFORE the W warm-up steps, so that the K timed steps run at the clock the part "
                         "SUSTAINS under this load rather than at a cold burst clock (default: 2 s on the headline workload, else 0)")
    ap.add_argument("--cpu-seconds", type=float, default=None,
                    help="wall budget of the cpu_baseline leg (0 = skip; default: 24 s at the headline shape, skipped otherwise)")
    ap.add_argument("--secondary", type=int, default=None,
                    help="1: also time the other BASELINE shapes (default at one GPU on the headline workload), 0: skip")
    ap.add_argument("--secondary-only", default=None,
                    help="comma-separated names of secondary entries (or 'all'): time ONLY those (no headline workload, no CPU leg) and "
                         "print {\"secondary\": ...}; tools/collect_profiles.sh profiles them one by one this way")
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default=None,
                    help="make one of the secondary BASELINE shapes the measured workload; c4 is BASELINE's STRONG-scaling "
                         "case: a fixed batch of 8192 regressors sharded over the ranks (8192 / N per GPU)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="fixed TOTAL number of regressors, sharded over the ranks in contiguous blocks (strong scaling); "
                         "default: --batch per GPU (weak scaling), except --config c4")
    ap.add_argument("--comm", choices=["rccl", "torch"], default="rccl",
                    help="N > 1: the evidence exchange through the library's own RCCL binding (blr_comm_init + "
                         "blr_logpdf_allgather_sum; torch.distributed only ships the unique id) or through torch.distributed")
    args = ap.parse_args()
    Din = None
    if args.config == "c3":
        args.D, args.N, args.dtype, args.noise, args.batch = 1024, 65536, "f32", "diagonal", 1
    elif args.config == "c4":
        args.D, args.N, args.dtype, args.noise, args.batch = 64, 1024, "f64", "isotropic", 8192
        if args.global_batch is None:
            args.global_batch = 8192
    elif args.config == "c5":
        args.D, args.N, args.dtype, args.noise, args.batch, Din = 2048, 16384, "f32", "isotropic", 1, 8
    headline = (args.D, args.N, args.dtype, args.noise, Din) == (128, 4096, "f64", "isotropic", None)
    if args.cpu_seconds is None:
        args.cpu_seconds = 24.0 if headline else 0.0
    if args.secondary_only:
        args.cpu_seconds = 0.0
    if args.preheat_seconds is None:
        args.preheat_seconds = 2.0 if headline else 0.0

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.secondary_only:
        # run bare (no launcher): become the launcher -- N fresh rank processes, BEFORE anything here touches the GPU -- and relay
        # rank 0's output and the job's exit code.  Never a silent one-rank run labelled n_gpus = 1.
        raise SystemExit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if args.secondary is None:
        args.secondary = 1 if (headline and world == 1) else 0
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

    import blr_amd  # noqa: F401
    from blr_amd import _abi

    h = _abi.Handle(local_rank)  # raises if the HIP extension or the GPU is missing: no fallback
    if os.environ.get("BLR_BENCH_SIDE_STREAM"):  # experiments: a non-default torch stream instead of the null stream
        torch.cuda.set_stream(torch.cuda.Stream(dev))
    stream = torch.cuda.current_stream(dev)
    h.set_stream(stream.cuda_stream)  # 0 = the HIP null stream = torch's default stream
    h.set_async(True)

    if args.secondary_only:
        only = None if args.secondary_only == "all" else set(args.secondary_only.split(","))
        sec = run_secondary(torch, _abi, h, dev, stream, only)
        print(json.dumps({"secondary": sec}), flush=True)  # (last line: the full records; tools/summarise_profiles.py reads it)
        h.close()
        return
    D, N = args.D, args.N
    strong = args.global_batch is not None
    if strong:  # contiguous blocks of a fixed batch (sharding.shard_range): the partition north_star names for config 4
        from blr_amd import sharding

        lo, hi = sharding.shard_range(args.global_batch, rank, world)
        B = hi - lo
        Bmax = max(sharding.shard_sizes(args.global_batch, world))  # collectives need equal counts: blocks are zero-padded
        global_batch = args.global_batch
    else:
        B = Bmax = args.batch
        global_batch = B * world
    cfg = {(128, 4096, "f64"): "c2", (1024, 65536, "f32"): "c3", (64, 1024, "f64"): "c4", (2048, 16384, "f32"): "c5"}
    wl = Workload(torch, _abi, h, dev, cfg.get((D, N, args.dtype), "custom"), B, D, N, args.dtype, args.noise,
                  123456 + 1 + (0 if strong else rank), Din, block=(lo, hi, 1024) if strong else None)
    lp_sum = torch.zeros((1,), dtype=torch.float64, device=dev)
    lp_loc = torch.zeros((Bmax,), dtype=torch.float64, device=dev)  # this rank's evidences (+ zero padding of an uneven block)
    wl.lp = lp_loc[:B]
    lp_all = torch.empty((Bmax * world,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)

    # N > 1: the exchange runs on the LIBRARY's RCCL binding (the path a host without torch would use); torch.distributed
    # only carries the 128-byte unique id, the barriers around the timed region and the max-over-ranks of the elapsed time
    # (BLR_BENCH_FORCE_COMM=1: run that code path with a one-rank communicator on a one-GPU box)
    use_lib_comm = (dist is not None and backend == "nccl" and args.comm == "rccl") or (
        world == 1 and os.environ.get("BLR_BENCH_FORCE_COMM") == "1")
    # The exchange runs OFF the critical path: a second handle carries the communicator on a side stream, step k's all-gather + sum
    # overlap step k + 1's launch (two alternating evidence buffers; events order the two streams).  Per step the job still produces
    # the total log evidence of that step's batch -- one step later.  (the exchange is latency-bound, 8 KB per rank; on
    # the launch stream it cost 30-50 us of a 150 us step at 8 GPUs.)
    hc, side = None, None
    comm_note = None
    if use_lib_comm:
        # The library's binding has only ever met a one-rank communicator on the build boxes.  A failure to set it up -- on ANY
        # rank -- must not cost the job its line: every rank learns of it (MIN over the ranks of a success flag, on the torch
        # process group that is already up) and all of them take the torch.distributed exchange instead, said in `config.sharding`.
        # BLR_BENCH_FAIL_LIB_COMM=1 forces that path (tests).
        ok, why = 1, ""
        try:
            if os.environ.get("BLR_BENCH_FAIL_LIB_COMM") == "1":
                raise RuntimeError("BLR_BENCH_FAIL_LIB_COMM=1")
            box = [_abi.Handle.comm_unique_id() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001
            ok, why, box = 0, f"{type(e).__name__}: {e}", [None]
        if dist is not None:
            dist.broadcast_object_list(box, src=0)
        if ok and box[0] is not None:
            try:
                hc = _abi.Handle(local_rank)
                side = torch.cuda.Stream(dev)
                hc.set_stream(side.cuda_stream)
                hc.set_async(True)
                hc.comm_init(world, rank, box[0])
            except Exception as e:  # noqa: BLE001
                ok, why = 0, f"{type(e).__name__}: {e}"
        else:
            ok = 0
        if dist is not None:
            flag = torch.tensor([ok], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok_all = int(flag.item())
        else:
            ok_all = ok
        if not ok_all:
            if hc is not None:
                try:
                    if ok:
                        hc.comm_destroy()
                    hc.close()
                except Exception:  # noqa: BLE001
                    pass
            hc, side, use_lib_comm = None, None, False
            comm_note = "library RCCL binding failed" + (f" here ({why})" if why else " on another rank") + \
                        ": exchange through torch.distributed instead"
            print(f"bench.py[rank {rank}]: {comm_note}", file=sys.stderr, flush=True)
    if use_lib_comm:
        lp_bufs = [lp_loc, torch.zeros_like(lp_loc)]
        lp_alls = [lp_all, torch.empty_like(lp_all)]
        lp_sums = [lp_sum, torch.zeros_like(lp_sum)]
        ev_done = [torch.cuda.Event(), torch.cuda.Event()]
        ev_free = [torch.cuda.Event(), torch.cuda.Event()]
        for e in ev_free:
            e.record(side)
    nstep = [0]

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        if use_lib_comm:
            k = nstep[0] & 1
            nstep[0] += 1
            stream.wait_event(ev_free[k])      # the exchange two steps back has read this buffer
            wl.lp = lp_bufs[k][:B]
        wl.launch()
        if ev is not None:
            ev[1].record(stream)
        # the path's only exchange: all-gather of the per-regressor log evidences (8 B each), then the SAME
        # fixed-order device sum on every rank -> identical bits for every rank count that divides the batch (SURVEY.md 8e;
        # uneven blocks are zero-padded, which regroups the sum: equal to rounding only)
        if use_lib_comm:  # ncclAllGather on the side stream + the fixed-order sum, one library call
            ev_done[k].record(stream)
            side.wait_event(ev_done[k])
            hc.logpdf_allgather_sum(Bmax, lp_bufs[k].data_ptr(), lp_alls[k].data_ptr(), lp_sums[k].data_ptr())
            ev_free[k].record(side)
        elif dist is not None:
            if backend == "nccl":
                dist.all_gather_into_tensor(lp_all, lp_loc)
            else:  # host-staged collective (validation only)
                host = torch.empty(Bmax * world, dtype=torch.float64)
                dist.all_gather_into_tensor(host, lp_loc.cpu())
                lp_all.copy_(host)
            h.logpdf_sum(_abi.MEM_DEVICE, Bmax * world, lp_all.data_ptr(), lp_sum.data_ptr())
        else:
            h.logpdf_sum(_abi.MEM_DEVICE, B, wl.lp.data_ptr(), lp_sum.data_ptr())

    # pre-heat: untimed steps for --preheat-seconds of device time (DVFS settles; MI355X_MICROARCH.md, clocks), then the contract's W
    # warm-up steps and EXACTLY K timed steps
    # (every step issues the exchange, a collective: all ranks must run the SAME number of pre-heat steps -- a loop that ends on
    # each rank's own clock lets two ranks straddle the boundary and hang.  One timed batch of 8, its time MAXed over the ranks,
    # gives the count every rank then runs.)
    preheat_steps = 0
    if args.preheat_seconds > 0:
        step()
        torch.cuda.synchronize(dev)
        t_pre = time.perf_counter()
        for _ in range(8):
            step()
        torch.cuda.synchronize(dev)
        t_batch = torch.tensor([time.perf_counter() - t_pre], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if dist is not None:
            dist.all_reduce(t_batch, op=dist.ReduceOp.MAX)
        n_batches = max(0, min(100000, int(np.ceil(args.preheat_seconds / max(float(t_batch.item()), 1e-6))) - 1))
        preheat_steps = 8
        for _ in range(n_batches):
            for _ in range(8):
                step()
            torch.cuda.synchronize(dev)
            preheat_steps += 8
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
    kern_all = [a.elapsed_time(b) for a, b in events]
    kern_ms = float(np.mean(kern_all))

    # sanity: the timed work produced valid results
    assert int(wl.info.abs().sum().item()) == 0, "a regressor failed to factorise"
    if use_lib_comm:
        lp_sum = lp_sums[(nstep[0] - 1) & 1]  # the last step's exchange
    total_evidence = float(lp_sum.item())
    assert np.isfinite(total_evidence)

    out = None
    if rank == 0:
        value = global_batch * args.steps / elapsed
        roof = wl.roofline(kern_ms)
        per_update, src = (None, None)
        if headline:
            per_update, src = pmc_traffic_per_update("c2_fused_i8_kernel_hbm" if roof["kernel"] == "fused_i8_kernel" else "c2_fused_small_kernel_hbm")
        peak_meas, peak_src = measured_matrix_peak(args.dtype)
        roof.update({
            "kernel_ms_min": float(np.min(kern_all)), "kernel_ms_median": float(np.median(kern_all)),
            "traffic": per_update * B if per_update else None,
            "traffic_source": src,
            "matrix_peak_measured_TFLOPps": peak_meas,
            "matrix_peak_measured_source": peak_src,
            "mfma_frac_of_measured_peak": (roof["mfma_frac"] * PEAK_TF[args.dtype] / peak_meas) if peak_meas else None,
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
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{wl.name}: independent regressors D={D}, N={N}, ColVecs, {args.noise} noise, {args.dtype}, "
                            f"Lw=I, fused posterior+logpdf" + (f", random-Fourier basis D_in={Din} on the device" if Din else ""),
                "D": D, "N": N, "batch_per_gpu": B, "global_batch": global_batch,
                "sharding": f"regressors x{world} ({'fixed batch, contiguous blocks' if strong else 'fixed block per GPU'}), no data-path "
                            f"collective; one all-gather of {Bmax * world} doubles via "
                            f"{'library RCCL (blr_logpdf_allgather_sum) on a side stream, overlapping the next step' if use_lib_comm else ('torch.distributed ' + backend) if dist is not None else 'nothing (one rank)'}"
                            + (f" [{comm_note}]" if comm_note else ""),
            },
            "roofline": roof,
            "total_log_evidence": total_evidence,
            "preheat_s": args.preheat_seconds,
        }
        if cpu_leg is not None:
            out["cpu_baseline"] = cpu_leg

    # ---- the other BASELINE shapes and the other hot-path rows (mean / var / rand / gradient / ...), same process, same timing
    # method (one GPU only): one short line each BEFORE the headline, full records in a side file
    if args.secondary and world == 1 and rank == 0:
        del wl, lp_all
        torch.cuda.empty_cache()
        sec = run_secondary(torch, _abi, h, dev, stream)
        out["n_secondary"] = len(sec)
        try:
            side = os.path.join(ROOT, "gpurun_out")
            os.makedirs(side, exist_ok=True)
            with open(os.path.join(side, "bench_secondary_latest.json"), "w") as f:
                json.dump({"headline": out, "secondary": sec}, f, indent=1)
            out["secondary_file"] = "gpurun_out/bench_secondary_latest.json"
        except OSError:
            out["secondary_file"] = None
    if use_lib_comm:
        hc.comm_destroy()
        hc.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    h.close()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which is flushed at exit -- AFTER anything Python has printed: drain it
        # first, so that the headline really is the LAST line of stdout
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(headline_line(out), flush=True)


if __name__ == "__main__":
    main()
