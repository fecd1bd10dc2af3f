"""Turns gpurun_out/profiles_raw/ (tools/collect_profiles.sh) into the committed profiles/ summaries of one round.
    python tools/summarise_profiles.py r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
RAW = "gpurun_out/profiles_raw"
DST = "profiles"
os.makedirs(DST, exist_ok=True)


def one(pattern):
    g = glob.glob(pattern)  # gpurun merges into gpurun_out/: an earlier collection's files may still sit next to the new ones
    return max(g, key=os.path.getmtime) if g else None


for name in ("bench_c2_f64", "bench_c2_f64_secondary", "bench_c2_f32", "bench_c3_f32", "bench_c5_f32", "bench_c4_f64", "bench_c3_f32_B8", "bench_c2_f64_fp64kernel"):
    src = os.path.join(RAW, name + ".json")
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(DST, f"{tag}_{name}.json"))
for cfg in ("c2", "c3", "c5", "c4", "c3b8", "marginals_var_c2_f64", "marginals_var_c3_f32", "rand_c2_f64_S64", "rand_c3_f32_S64", "logpdf_grad_c2_f64",
            "marginals_var_D512_B16_f32", "logpdf_multi_c3_f32_S64"):
    f = one(f"{RAW}/stats_{cfg}/*/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(DST, f"{tag}_{cfg}_kernel_stats.csv"))


def durations(dirname, kernel_substr):
    """min / median / mean duration (us) of a kernel's dispatches in a `rocprofv3 --kernel-trace --stats` run (one
    outlier launch moves the mean by 3 %)"""
    f = one(f"{RAW}/{dirname}/*/*_kernel_trace.csv")
    if not f:
        return None
    by_grid = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kernel_substr in r["Kernel_Name"]:
            by_grid[int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if not by_grid:
        return None
    out = {"kernel": kernel_substr, "launch_shapes": {}}
    step_med = step_mean = 0.0
    for g, d in sorted(by_grid.items()):
        d.sort()
        out["launch_shapes"][str(g)] = {"calls": len(d), "min_us": d[0], "median_us": d[len(d) // 2], "mean_us": sum(d) / len(d), "max_us": d[-1]}
        step_med += d[len(d) // 2]
        step_mean += sum(d) / len(d)
    # (the int8 route launches the kernel twice per call above 1024 regressors: a probe slice of 256, then the rest)
    out["per_step_median_us"], out["per_step_mean_us"] = step_med, step_mean
    return out


def counters(dirname, kernel_substr):
    f_new = one(f"{RAW}/{dirname}/*/*_counter_collection.csv")
    if not f_new:
        return None
    fs = [f_new]
    ts = [f_new.replace("_counter_collection.csv", "_kernel_trace.csv")]
    if not os.path.exists(ts[0]):
        return None
    dur = {}
    for t in ts:  # one file per traced process; dispatch ids are unique within a process directory
        key = os.path.dirname(t)
        for r in csv.DictReader(open(t)):
            dur[(key, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    by = collections.defaultdict(dict)
    for f in fs:
        key = os.path.dirname(f)
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                by[(key, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
                by[(key, r["Dispatch_Id"])]["_grid"] = int(r["Grid_Size"])
    if not by:
        return None
    gmax = max(c["_grid"] for c in by.values())  # the dominant launch shape
    sel = {d: c for d, c in by.items() if c["_grid"] == gmax and d in dur}
    if not sel:
        return None
    out = {"dispatches": len(sel), "grid": gmax, "avg_duration_ns": sum(dur[d] for d in sel) / len(sel)}
    names = sorted({k for c in sel.values() for k in c if not k.startswith("_")})
    for k in names:
        out[k] = sum(c[k] for c in sel.values()) / len(sel)
    return out


kd = {}
for cfg, kern in (("c2", "fused_i8_kernel"), ("c3", "gram_planes"), ("c5", "gram_planes"), ("c4", "fused_wave_kernel"), ("c3b8", "gram_planes"),
                  ("marginals_var_c2_f64", "marginals_gemm_kernel"), ("marginals_var_c3_f32", "marg_blocksub_kernel"),
                  ("logpdf_grad_c2_f64", "grad_gemm_kernel")):
    r = durations(f"stats_{cfg}", kern)
    if r:
        kd[cfg] = r
if kd:
    json.dump(kd, open(os.path.join(DST, f"{tag}_kernel_durations.json"), "w"), indent=1)

summary = {"note": "per-dispatch means for the dominant kernel; rocprofv3 --pmc, one counter group per pass "
                   "(tools/collect_profiles.sh); FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x2 on gfx950 "
                   "(MI355X_MICROARCH.md, HBM/rocprofv3 section)"}
fetch = counters("pmc_FETCH_SIZE_c2fp64", "fused_small_kernel")  # (the fp64 kernel: runs under BLR_MI355X_NO_I8_GRAM)
write = counters("pmc_WRITE_SIZE_c2fp64", "fused_small_kernel")
sq2 = counters("pmc_sq_c2fp64", "fused_small_kernel")
sq3 = counters("pmc_sq_c3", "gram_planes")
if fetch and write:
    rd = fetch["FETCH_SIZE"] * 1024.0 * 2.0
    wr = write["WRITE_SIZE"] * 1024.0
    summary["c2_fused_small_kernel_hbm"] = {"FETCH_SIZE_KiB": fetch["FETCH_SIZE"], "WRITE_SIZE_KiB": write["WRITE_SIZE"],
                                            "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                                            "hbm_bytes_per_launch": rd + wr, "units_per_launch": 4096}
sq2f = counters("pmc_sq_c2f32", "fused_small_kernel")
f4 = counters("pmc_fetch_c4", "fused_wave_kernel")
w4 = counters("pmc_write_c4", "fused_wave_kernel")
if f4 and w4:
    rd = f4["FETCH_SIZE"] * 1024.0 * 2.0
    wr = w4["WRITE_SIZE"] * 1024.0
    summary["c4_fused_wave_kernel_hbm"] = {"FETCH_SIZE_KiB": f4["FETCH_SIZE"], "WRITE_SIZE_KiB": w4["WRITE_SIZE"],
                                            "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                                            "hbm_bytes_per_launch": rd + wr, "units_per_launch": 8192}
sq4 = counters("pmc_sq_c4", "fused_wave_kernel")
sq5 = counters("pmc_sq_c5", "gram_planes")
sq3b8 = counters("pmc_sq_c3b8", "gram_planes")  # 8 regressors of c3's shape in one launch
for key, d, kern, units in (("c2_f32_fused_small_kernel_hbm", "c2f32", "fused_small_kernel", 4096), ("c3_gram_planes_kernel_hbm", "c3", "gram_planes", 1),
                            ("c5_gram_planes_kernel_hbm", "c5", "gram_planes", 1)):
    fe, wr_ = counters(f"pmc_fetch_{d}", kern), counters(f"pmc_write_{d}", kern)
    if fe and wr_:
        rd = fe["FETCH_SIZE"] * 1024.0 * 2.0
        wr = wr_["WRITE_SIZE"] * 1024.0
        summary[key] = {"FETCH_SIZE_KiB": fe["FETCH_SIZE"], "WRITE_SIZE_KiB": wr_["WRITE_SIZE"], "hbm_read_bytes_per_launch": rd,
                        "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "units_per_launch": units,
                        "kernel": kern, "avg_duration_ns": fe["avg_duration_ns"]}
for src in sorted(glob.glob(os.path.join(os.path.dirname(RAW), "microbench", "*.txt"))):  # tools/run_microbench.sh
    if os.path.getsize(src):
        shutil.copy(src, os.path.join(DST, f"{tag}_microbench_{os.path.basename(src)}"))
for extra in ("ring_probe.txt", "power_probe.txt", "i8_gram.txt", "i8_sustained.txt", "c4_clocks.txt", "chol_bench.txt", "bf16_probe.txt", "marg_bench.txt", "marg128_bench.txt", "group_scan.txt", "layout_time.txt", "rowvecs_gap.txt"):
    src = os.path.join(RAW, extra)
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(DST, f"{tag}_microbench_{extra}"))
for key, c in (("c2_fused_small_kernel_sq", sq2), ("c2_f32_fused_small_kernel_sq", sq2f), ("c3_gram_planes_kernel_sq", sq3),
               ("c4_fused_wave_kernel_sq", sq4), ("c5_gram_planes_kernel_sq", sq5), ("c3_B8_gram_planes_kernel_sq", sq3b8)):
    if c:
        ns = c["avg_duration_ns"]
        c["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / ns
        c["mfma_busy_fraction_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["effective_clock_GHz"] * ns)
        summary[key] = c


def call_bytes(dirname, counter):
    """sum of a counter over every blr:: dispatch of a `bench.py --secondary-only <entry>` run"""
    f_new = one(f"{RAW}/{dirname}/*/*_counter_collection.csv")
    if not f_new:
        return None
    tot = 0.0
    for r in csv.DictReader(open(f_new)):
        if "blr::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
    return tot


for f in sorted(glob.glob(f"{RAW}/sec_fetch_*.json")):
    e = os.path.basename(f)[len("sec_fetch_"):-len(".json")]
    try:
        calls = json.loads(open(f).read().strip().splitlines()[-1])["secondary"][e]["calls"]  # (last stdout line = the full records)
    except Exception:
        continue
    fe, wr = call_bytes(f"sec_fetch_{e}", "FETCH_SIZE"), call_bytes(f"sec_write_{e}", "WRITE_SIZE")
    if fe is None or wr is None:
        continue
    rd, wb = fe * 1024.0 * 2.0 / calls, wr * 1024.0 / calls  # (FETCH_SIZE x 2 on gfx950, as above)
    summary[e + "_hbm"] = {"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wb, "hbm_bytes_per_launch": rd + wb,
                           "units_per_launch": 1, "calls_profiled": calls,
                           "note": "all blr:: kernels of one call (bench.py --secondary-only), warm-up calls included in the mean"}
sq_i8 = counters("pmc_sq_c2_i8", "fused_i8_kernel")
if sq_i8:
    ns = sq_i8["avg_duration_ns"]
    sq_i8["effective_clock_GHz"] = sq_i8["GRBM_GUI_ACTIVE"] / 8.0 / ns
    sq_i8["mfma_busy_fraction_of_simd_cycles"] = sq_i8["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * sq_i8["effective_clock_GHz"] * ns)
    summary["c2_fused_i8_kernel_sq"] = sq_i8
for key, d in (("c2_fused_i8_kernel_hbm", "c2"),):
    fe, wr_ = counters(f"pmc_fetch_{d}", "fused_i8_kernel"), counters(f"pmc_write_{d}", "fused_i8_kernel")
    if fe and wr_:
        rd = fe["FETCH_SIZE"] * 1024.0 * 2.0
        wr = wr_["WRITE_SIZE"] * 1024.0
        summary[key] = {"FETCH_SIZE_KiB": fe["FETCH_SIZE"], "WRITE_SIZE_KiB": wr_["WRITE_SIZE"], "hbm_read_bytes_per_launch": rd,
                        "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "units_per_launch": fe["grid"] // 512,
                        "kernel": "fused_i8_kernel", "avg_duration_ns": fe["avg_duration_ns"],
                        "note": "the dominant launch of a call: above 1024 regressors the int8 route runs a probe slice of 256 workgroups "
                                "first; this is the launch of the remaining ones (one 512-thread workgroup per regressor)"}
if "c2_fused_small_kernel_hbm" in summary:  # the A/B secondary entry of the driver-line workload on the fp64 kernel
    summary["c2_f64_fp64_kernel_hbm"] = dict(summary["c2_fused_small_kernel_hbm"], units_per_launch=1,
                                             note="= c2_fused_small_kernel_hbm (one launch per call)")
# what this collection measured: the kernel sources as they stood (bench.py marks a `traffic` figure stale when they have changed since)
import hashlib
import subprocess
src_dir = os.path.join("bayesianlinearregressors.jl_amd", "csrc")
summary["_kernel_source_sha16"] = {f: hashlib.sha256(open(os.path.join(src_dir, f), "rb").read()).hexdigest()[:16]
                                   for f in sorted(os.listdir(src_dir)) if f.endswith((".hpp", ".hip"))}
try:
    summary["_collected_at_commit"] = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    pass
json.dump(summary, open(os.path.join(DST, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
