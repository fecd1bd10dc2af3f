"""CPU-side checks: the C-ABI library loads and exports every symbol include/blr_mi355x.h declares,
the product path fails loudly without a GPU (no fallback), and the host-side index/shape logic
(the x_as_colvecs mirror, reference src/bayesian_linear_regression.jl:20-31) is exact."""
import os
import re

import numpy as np
import pytest

import blr_amd
from blr_amd import _abi
from blr_amd import regressor as R


def _declared_functions(repo_root):
    text = open(os.path.join(repo_root, "include", "blr_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(blr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(repo_root):
    lib = _abi.load_library()
    names = _declared_functions(repo_root)
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/blr_mi355x.h but not exported"
    assert set(names) == set(_abi.EXPORTED_SYMBOLS), "ctypes binding and header disagree"
    assert lib.blr_abi_version() == 1
    assert lib.blr_last_error(None) == b"null handle"


def test_no_cpu_fallback_without_gpu():
    lib = _abi.load_library()
    if lib.blr_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_abi.BLRError):
        _abi.Handle(0)
    f = blr_amd.BayesianLinearRegressor(np.zeros(2), blr_amd.Diagonal(np.ones(2)))
    with pytest.raises(_abi.BLRError):
        blr_amd.logpdf(f(np.zeros((2, 3)), 0.1), np.zeros(3))
    import subprocess
    import sys

    # the product package never imports the oracle
    code = "import sys, blr_amd; assert not any(m.startswith('oracle') for m in sys.modules), 'oracle imported'"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=os.path.dirname(os.path.dirname(__file__)))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", str(tmp_path / "libblr_mi355x.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _abi.load_library()


def test_x_layout_is_zero_copy_and_bit_exact():
    D, N = 3, 5
    X = np.arange(D * N, dtype=np.float64).reshape(D, N)

    def element(arr, layout, ld, d, n):
        flat = arr.ravel(order="K")  # memory order
        return flat[d + n * ld] if layout == _abi.LAYOUT_COLVECS else flat[n + d * ld]

    cases = [
        (np.asfortranarray(X), _abi.LAYOUT_COLVECS, D),  # raw D x N matrix -> ColVecs (AbstractGPs)
        (np.ascontiguousarray(X), _abi.LAYOUT_ROWVECS, N),
        (R.ColVecs(np.asfortranarray(X)), _abi.LAYOUT_COLVECS, D),
        (R.ColVecs(np.ascontiguousarray(X)), _abi.LAYOUT_ROWVECS, N),
        (R.RowVecs(np.ascontiguousarray(X.T)), _abi.LAYOUT_COLVECS, D),  # reference :24, lazy adjoint
        (R.RowVecs(np.asfortranarray(X.T)), _abi.LAYOUT_ROWVECS, N),
    ]
    for x, want_layout, want_ld in cases:
        arr, layout, ld, d_, n_ = R._x_layout(x, np.float64)
        assert (layout, ld, d_, n_) == (want_layout, want_ld, D, N)
        src = x.X if hasattr(x, "X") else x
        assert np.shares_memory(arr, src), "layout normalisation must not copy"
        for d in range(D):
            for n in range(N):
                assert element(arr, layout, ld, d, n) == X[d, n]
    with pytest.raises(TypeError, match="ColVecs or RowVecs"):  # reference :26-31
        R._x_layout([np.zeros(3), np.zeros(3)], np.float64)
    # non-contiguous views are compacted once
    arr, layout, ld, d_, n_ = R._x_layout(R.ColVecs(np.zeros((6, 10))[::2, ::2]), np.float64)
    assert (d_, n_) == (3, 5) and (arr.flags.c_contiguous or arr.flags.f_contiguous)


def test_noise_and_prior_classification():
    s, kind = R._noise(0.1, 7, np.float64)
    assert kind == _abi.NOISE_ISOTROPIC and s.shape == (1,) and s[0] == 0.1
    s, kind = R._noise(R.Diagonal(np.ones(7)), 7, np.float32)
    assert kind == _abi.NOISE_DIAGONAL and s.dtype == np.float32
    with pytest.raises(ValueError):
        R._noise(np.ones(6), 7, np.float64)
    s, kind = R._noise(np.eye(7), 7, np.float64)  # dense Sigma_y: whitened on the device (blr_posterior_dense_noise_*)
    assert kind == _abi.NOISE_DENSE and s.flags.f_contiguous and s.shape == (7, 7)
    with pytest.raises(ValueError):
        R._noise(np.eye(6), 7, np.float64)
    # the mirror checks positivity where the reference factorises and the kernel behind the call does not report it
    with pytest.raises(_abi.PosDefException) as ei:
        R._noise(np.array([1.0, 2.0, 0.0, -1.0, 1.0, 1.0, 1.0]), 7, np.float64, need_cholesky=True)
    assert ei.value.info == 3
    with pytest.raises(ValueError):  # length(mw) != input dimension: never a read past the end of mw
        R._mean_vector(np.zeros(3), 5, np.float64)
    A = np.array([[2.0, 1.0], [1.0, 3.0]])
    for Lw, want in ((A, _abi.PRIOR_DENSE), (R.Symmetric(A), _abi.PRIOR_DENSE), (R.PDMat(np.linalg.cholesky(A).T),
                                                                                 _abi.PRIOR_UPPER_FACTOR)):
        arr, kind, ldl = R._prior(Lw, 2, np.float64)
        assert kind == want and ldl == 2 and arr.flags.f_contiguous
    arr, kind, ldl = R._prior(R.Diagonal(np.ones(2)), 2, np.float64)
    assert kind == _abi.PRIOR_DIAGONAL
    with pytest.raises(ValueError):
        R._prior(np.eye(3), 2, np.float64)
    np.testing.assert_allclose(R.PDMat(np.linalg.cholesky(A).T).toarray(), A)
    np.testing.assert_allclose(R.Symmetric(np.triu(A)).toarray(), A)


def test_wrapper_type_closure_and_exports():
    # reference :92-93 and src/BayesianLinearRegressors.jl:11-12
    T = np.triu(np.ones((2, 2)))
    A = T.T @ T
    assert isinstance(R._wrap_like(R.PDMat(T), T, None), R.PDMat)
    assert isinstance(R._wrap_like(np.eye(2), T, A), R.Symmetric)
    assert isinstance(R._wrap_like(R.Diagonal(np.ones(2)), T, A), R.Symmetric)
    for name in ("logpdf", "rand", "mean", "std", "cov", "BayesianLinearRegressor", "marginals", "posterior",
                 "BasisFunctionRegressor"):
        assert hasattr(blr_amd, name)
    f = blr_amd.BayesianLinearRegressor(np.zeros(2), np.eye(2))
    fx = f(np.zeros((2, 3)))
    assert fx.Sy == 1e-18 and fx.f is f  # AbstractGPs default noise
    bf = blr_amd.BasisFunctionRegressor(f, lambda x: x)
    assert R._to_finite_blr(bf(np.zeros((2, 3)), 0.5)).f is f
    Z = R._randn(np.random.default_rng(0), 3, 4, np.float64)
    assert Z.shape == (3, 4) and Z.flags.f_contiguous
    np.testing.assert_array_equal(Z.ravel(order="F"), np.random.default_rng(0).standard_normal(12))


def test_hot_kernels_keep_two_workgroups_per_cu(tmp_path):
    """Regression guard (measured 475 k -> 351 k updates/s): the noinline phase functions are compiled once for all
    their callers, so ONE caller without the 2-waves-per-SIMD launch bound makes hipcc budget them for 512 registers
    and silently halves the occupancy of the fused kernel.  Read the register counts from the code object."""
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("ROCm LLVM tools not installed")
    so = shutil.copy(_abi.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    cos = [p for p in os.listdir(tmp_path) if "gfx950" in p]
    assert cos, "no gfx950 code object in the library"
    # (one code object per translation unit: the host ABI with most kernels, the int8-sliced kernel once per noise kind)
    notes = "".join(subprocess.run([readelf, "--notes", str(tmp_path / c)], check=True, capture_output=True, text=True).stdout for c in sorted(cos))
    kernels = {}
    cur = {}
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(agpr_count|vgpr_count|name):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "agpr_count" and cur:
            cur = {}
        cur[m.group(1)] = m.group(2)
        if len(cur) == 3:
            kernels[cur["name"]] = int(cur["vgpr_count"]) + 0  # .vgpr_count already includes the AGPRs on gfx950
            cur = {}
    hot = {k: v for k, v in kernels.items() if "fused_small_kernel" in k or "gram_tile_kernel" in k}
    assert len(hot) >= 60, f"expected the fused/gram kernels in the code object, found {len(hot)}"
    over = {k: v for k, v in hot.items() if v > 256}
    assert not over, f"kernels above 256 registers (1 workgroup per CU): {over}"
    i8 = {k: v for k, v in kernels.items() if "fused_i8_kernel" in k}
    n_i8 = 1 if "asan" in os.path.basename(str(_abi.LIB_PATH)) else 8  # (two noise kinds x two layouts x two digit-group plans; the sanitizer build carries one form: BLR_DEV_FAST)
    assert len(i8) == n_i8 and max(i8.values()) <= 256, f"the eight forms of the int8-sliced kernel (8 waves of 256 registers): {i8}"
    # Scratch of the hot kernels.  The phase functions have internal linkage and no tail-called call site, so LLVM's
    # interprocedural register allocation drops their callee-saved-register saves (blr_fused_small.hpp, BLR_PHASE); what is left is
    # the few values a kernel keeps across its calls, which the code object now books as the KERNEL's spills.  The honest measure is
    # the scratch per lane: 612 B (fp64 fallback kernel) and 320 B (int8 kernel) before; the fallback kernel's own glue code
    # between the phases is a set of phases too, so the kernel keeps two scalar values across its calls (76 B, 2 vector spills).
    scratch, spills, name = {}, {}, None
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(name|private_segment_fixed_size|vgpr_spill_count):\s+(\S+)", line)
        if m and m.group(1) == "name":
            name = m.group(2)
        elif m and name is not None:
            (scratch if m.group(1) == "private_segment_fixed_size" else spills)[name] = int(m.group(2))
    i8_scratch = {k: v for k, v in scratch.items() if "fused_i8_kernel" in k}
    assert len(i8_scratch) == n_i8
    assert max(i8_scratch.values()) <= 64, f"int8 kernels: bytes of scratch per lane {i8_scratch}"
    assert max(v for k, v in spills.items() if "fused_i8_kernel" in k) <= 8
    fb = {k: v for k, v in scratch.items() if "fused_small_kernelIdLi8ELi4" in k}
    assert len(fb) == 1 and max(fb.values()) <= 128, f"fp64 fallback kernel: bytes of scratch per lane {fb}"
    assert max(v for k, v in spills.items() if "fused_small_kernelIdLi8ELi4" in k) <= 4
    # config 4's one-wave kernel (VERDICT r4 #4a): its back substitution is a phase of its own, so the 128 lane masks of the unrolled
    # pivots are no longer hoisted out of the loop over the regressors and parked in VGPR lanes (348 scalar spills before)
    sg, name = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(name|sgpr_spill_count):\s+(\S+)", line)
        if m and m.group(1) == "name":
            name = m.group(2)
        elif m and name is not None:
            sg[name] = int(m.group(2))
    wave = {k: v for k, v in sg.items() if "fused_wave_kernelIdLi4ELi1" in k}
    assert len(wave) == 1 and max(wave.values()) <= 32, f"fused_wave_kernel<double, 4, 1>: scalar spills {wave}"


def test_julia_shim_ccall_signatures_match_the_header(repo_root):
    """julia/BLRMI355X.jl cannot run in this image (no julia): at least keep every ccall's argument-type tuple and argument
    list in step with the prototype in include/blr_mi355x.h (count and integer / pointer / floating kind per position)."""
    header = open(os.path.join(repo_root, "include", "blr_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(blr_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", header, flags=re.S):
        params = [p.strip() for p in m.group(2).replace("\n", " ").split(",") if p.strip() and p.strip() != "void"]
        kinds = []
        for p in params:
            if "*" in p:
                kinds.append("ptr")
            elif re.match(r"(const\s+)?(double|float)\b", p):
                kinds.append("fp")
            else:
                kinds.append("int")
        protos[m.group(1)] = kinds
    jl = open(os.path.join(repo_root, "julia", "BLRMI355X.jl")).read()

    def split_top(sarg):
        out, depth, cur = [], 0, ""
        for ch in sarg:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            if ch == "," and depth == 0:
                out.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            out.append(cur.strip())
        return out

    seen = 0
    for m in re.finditer(r"ccall\(\(:(blr_[a-z0-9_]+), LIB\),\s*(\w+),\s*\(", jl):
        name = m.group(1)
        # the type tuple: balanced parentheses starting at the match end - 1
        i = m.end() - 1
        depth, j = 0, i
        while True:
            depth += jl[j] == "("
            depth -= jl[j] == ")"
            j += 1
            if depth == 0:
                break
        types = split_top(jl[i + 1:j - 1])
        # the call arguments run to the parenthesis that closes ccall(
        k, depth = j, 1
        while depth:
            depth += jl[k] == "("
            depth -= jl[k] == ")"
            k += 1
        args = split_top(jl[j:k - 1].lstrip(", \n"))
        assert name in protos, f"{name} is not declared in the header"
        want = protos[name]
        assert len(types) == len(want), f"{name}: {len(types)} ccall types vs {len(want)} parameters in the header"
        assert len(args) == len(want), f"{name}: {len(args)} ccall arguments vs {len(want)} parameters in the header"
        for pos, (t, kind) in enumerate(zip(types, want)):
            got = "ptr" if t.startswith(("Ptr", "Ref", "Cstring")) else ("fp" if t in ("T", "Cdouble", "Cfloat", "Float64", "Float32") else "int")
            assert got == kind, f"{name}: argument {pos + 1} is {t} in the shim but {kind} in the header"
        seen += 1
    assert seen >= 34  # every entry point a Julia host needs, incl. dense noise, cov, batched device form and the RCCL exchange


def test_comm_entry_points_validate_without_a_gpu():
    """blr_comm_* / blr_logpdf_allgather_sum (RCCL called directly): argument checks need no device."""
    from blr_amd import _abi

    lib = _abi.load_library()
    assert lib.blr_comm_init(None, 2, 0, None) == -1
    assert lib.blr_comm_size(None) == -1 and lib.blr_comm_rank(None) == -1
    assert lib.blr_comm_destroy(None) == -1
    assert lib.blr_logpdf_allgather_sum(None, 4, None, None, None) == -1
    assert lib.blr_allreduce_sum(None, 1, None, 4) == -1
    assert lib.blr_comm_unique_id(None) == -1



def test_bench_headline_line_stays_under_3k(repo_root):
    # the driver keeps 8 KB of stdout and the round-4 line had grown to 21.6 KB (31 secondary entries), so the head of it
    # -- value, roofline, cpu_baseline -- was lost.  bench.py now prints the secondaries as one short line each BEFORE a compact
    # headline object; run the formatter on the recorded round-4 result (the widest one there is) and hold it to the budget.
    import importlib.util
    import json

    ROOT = repo_root
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_c2_f64.json")))
    rec["roofline"].update({"kernel_ms_min": 5.04, "kernel_ms_median": 5.2, "f64_equiv_frac": 0.68})
    rec["preheat_s"], rec["n_secondary"], rec["secondary_file"] = 2.0, len(rec["secondary"]), "gpurun_out/bench_secondary_latest.json"
    line = bench.headline_line(rec)
    assert len(line) < 3072 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline", "scaling",
              "vs_baseline", "higher_is_better", "data"):
        assert k in d, k
    assert "secondary" not in d
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac", "int8_frac", "f64_equiv_frac", "kernel"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert abs(d["value"] - rec["value"]) <= 1e-8 * rec["value"]
    for name, e in rec["secondary"].items():
        s = bench.secondary_line(name, e)
        assert len(s) <= 300 and json.loads(s)["secondary"] == name
    assert len(bench.secondary_line("x", {"error": "E" * 5000})) < 300


def test_bench_bare_multi_gpu_invocation_never_runs_one_rank(repo_root):
    # `python bench.py --gpus N` without a launcher must either run N ranks or fail -- never one rank labelled
    # n_gpus = 1.  The spawn happens before anything touches the GPU, so its failure modes are checkable here:
    #  (a) the launcher cannot start -> non-zero exit, no JSON headline;
    #  (b) a launcher-provided WORLD_SIZE that disagrees with --gpus -> non-zero exit (also for WORLD_SIZE = 1).
    import subprocess
    import sys

    bench = os.path.join(repo_root, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--cpu-seconds", "0"], env=dict(env, BLR_BENCH_LAUNCHER="blr_no_such_launcher_module"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not any(ln.startswith("{") and '"n_gpus"' in ln for ln in r.stdout.splitlines())
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--cpu-seconds", "0"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--cpu-seconds", "0"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
