#!/bin/bash
# Runs on the GPU box (gpurun): bench lines + rocprofv3 kernel stats + PMC passes for the headline workload (c2) and the
# secondary shapes.  Everything lands in gpurun_out/profiles_raw/; tools/summarise_profiles.py turns it into profiles/.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_raw
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-seconds 0 --secondary 0 --preheat-seconds 0"
# the driver's command: one short line per secondary entry, then the headline as the LAST line; full secondary records in a side file
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c2_f64.out 2> $OUT/bench_c2_f64.err
tail -1 $OUT/bench_c2_f64.out > $OUT/bench_c2_f64.json
cp $R/gpurun_out/bench_secondary_latest.json $OUT/bench_c2_f64_secondary.json 2>/dev/null
$B --dtype f32 > $OUT/bench_c2_f32.json 2>/dev/null
$B --config c3 --steps 20 --warmup 3 > $OUT/bench_c3_f32.json 2>/dev/null
$B --config c5 --steps 20 --warmup 3 > $OUT/bench_c5_f32.json 2>/dev/null
$B --config c4 > $OUT/bench_c4_f64.json 2>/dev/null
# (c2: the DRIVER's command, pre-heat included -- the average duration of fused_i8_kernel in this summary is the sustained one the
#  headline's HIP-event figure has to agree with; without the pre-heat the first cold launches pull it up by 3-5 %)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- $B --config c3 --steps 20 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- $B --config c5 --steps 20 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c4 -- $B --config c4 > /dev/null 2>&1
# 8 regressors of c3's shape in one call: every launch of the update covers the group (posterior_large_group)
$B --D 1024 --N 65536 --dtype f32 --noise diagonal --batch 8 --steps 10 --warmup 2 > $OUT/bench_c3_f32_B8.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3b8 -- $B --D 1024 --N 65536 --dtype f32 --noise diagonal --batch 8 --steps 10 --warmup 2 > /dev/null 2>&1
# PMC: separate passes (TCC FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only
SQ="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
# (the default c2 path is the int8-sliced Gram; the fp64 kernel of the same shape under BLR_MI355X_NO_I8_GRAM -- the variable is set
# for rocprofv3 itself, the program after `--` stays python3)
for c in FETCH_SIZE WRITE_SIZE; do
  BLR_MI355X_NO_I8_GRAM=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_c2fp64 -- $B --steps 5 --warmup 2 > /dev/null 2>&1
done
BLR_MI355X_NO_I8_GRAM=1 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_c2fp64 -- $B --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c2 -- $B --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c2 -- $B --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_c2f32 -- $B --dtype f32 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_c3 -- $B --config c3 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_c4 -- $B --config c4 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c4 -- $B --config c4 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c4 -- $B --config c4 --steps 5 --warmup 2 > /dev/null 2>&1
# HBM traffic of the secondary shapes: the Gram launch of c3 / c5, the fused kernel of c2 in f32
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c2f32 -- $B --dtype f32 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c2f32 -- $B --dtype f32 --steps 5 --warmup 2 > /dev/null 2>&1
for c in c3 c5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$c -- $B --config $c --steps 5 --warmup 2 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$c -- $B --config $c --steps 5 --warmup 2 > /dev/null 2>&1
done
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_c5 -- $B --config c5 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_c3b8 -- $B --D 1024 --N 65536 --dtype f32 --noise diagonal --batch 8 --steps 5 --warmup 2 > /dev/null 2>&1
# the clock the part holds: ring loop of the headline kernel on zero / random operands, cache-resident / streamed
( cd $R/tools && [ -x ./ring_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRING_MWZ=true -I$R/bayesianlinearregressors.jl_amd/csrc ring_probe.hip -o ring_probe 2>/dev/null
  { echo "# tools/ring_probe 8192 0 (zero operands)"; ./ring_probe 8192 0; echo "# tools/ring_probe 8192 1 (random operands)"; ./ring_probe 8192 1; } > $OUT/ring_probe.txt 2>&1
  python3 $R/tools/power_probe.py > $OUT/power_probe.txt 2>&1
  python3 $R/tools/group_scan.py > $OUT/group_scan.txt 2>&1
  python3 $R/tools/layout_time.py > $OUT/layout_time.txt 2>&1
  python3 $R/tools/rowvecs_gap.py > $OUT/rowvecs_gap.txt 2>&1
  { ./marg128_bench 64 4096 8; ./marg128_bench 64 4096 16; ./marg128_bench 256 4096 2; [ -x ./marg128_bench_st ] && ./marg128_bench_st 64 4096 8 1 | grep "image kernel" | head -1; } > $OUT/marg128_bench.txt 2>&1
  { ./marg_bench 1024 65536 20; [ -x ./marg_bench16 ] && ./marg_bench16 1024 65536 20; ./marg_bench 1024 999 20; [ -x ./marg_bench16 ] && ./marg_bench16 1024 999 20; [ -x ./marg_bench_st ] && ./marg_bench_st 1024 65536 5 | grep wave; } > $OUT/marg_bench.txt 2>&1
  # int8-sliced Gram against the fp64 kernel, same inputs: rates, agreement, the retry path (mode 1), per-phase cycle stamps
  [ -x ./i8_gram ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -I$R/bayesianlinearregressors.jl_amd/csrc i8_gram.hip -o i8_gram 2>/dev/null
  { for m in 0 2 4 1; do ./i8_gram 4096 4096 10 $m; done; ./i8_gram 512 4096 3 5; ./i8_gram 512 1024 3 0; ./i8_gram 256 16384 3 0; ./i8_gram 4096 4096 10 3; I8_MW=1 ./i8_gram 4096 4096 10 0;
    [ -x ./i8_gram_st ] && ./i8_gram_st 4096 4096 4 0 | grep "wave"; } > $OUT/i8_gram.txt 2>&1
  # config 4's one-wave kernel: whole / Gram phase only / MFMAs without the stream / stream without the MFMAs, with the clock each holds
  for e in 0 1 4 5; do [ -x ./fused_bench_w$e ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFB_WAVE -DFB_D=64 -DBLR_WAVE_CLK -DBLR_EXP=$e -I$R/bayesianlinearregressors.jl_amd/csrc fused_bench.hip -o fused_bench_w$e 2>/dev/null; done
  { echo "# tools/fused_bench, -DFB_WAVE -DFB_D=64 -DBLR_WAVE_CLK: EXP=0 the kernel, 1 Gram phase only, 4 no LDS-DMA (MFMAs on stale data), 5 LDS-DMA only"; for e in 0 1 4 5; do ./fused_bench_w$e 8192 1024 0 30 | tail -2; done; } > $OUT/c4_clocks.txt 2>&1
  { echo "# tools/chol_bench: phase_chol, cycles per section (wave 0), f32 and f64; tools/chol_bench 2: the experiments of blr_chol_dpp_experiment.hpp"; ./chol_bench; ./chol_bench 2; } > $OUT/chol_bench.txt 2>&1
  # bf16 matrix instructions next to the f32 one, and the unit check of the bf16 x 3 product the fp32 large-D Gram uses
  [ -x ./bf16_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 bf16_probe.hip -o bf16_probe 2>/dev/null
  [ -x ./bf3_unit ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 bf3_unit.hip -o bf3_unit 2>/dev/null
  { echo "# tools/bf16_probe: matrix instructions of gfx950, 8 accumulators per wave (the 16 x 16 forms are latency-bound at that depth)"; ./bf16_probe; echo "# tools/bf3_unit"; ./bf3_unit; } > $OUT/bf16_probe.txt 2>&1
  # sustained (>= 2.5 s of back-to-back launches) in-kernel clock of the int8 kernel, N(0,1) operands against zeros
  [ -x ./i8_gram_st ] && { I8_SUSTAINED=2.5 ./i8_gram_st 4096 4096 4 0 | grep "sustained"; I8_SUSTAINED=2.5 ./i8_gram_st 4096 4096 4 3 | grep "sustained"; } > $OUT/i8_sustained.txt 2>&1 )
# every secondary entry of the driver line: kernel stats of the hot-path rows, HBM bytes per CALL (all of a call's kernels) by PMC
for e in marginals_var_c2_f64 marginals_var_c3_f32 marginals_var_D512_B16_f32 rand_c2_f64_S64 rand_c3_f32_S64 logpdf_grad_c2_f64 logpdf_multi_c3_f32_S64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$e -- $B --secondary-only $e > /dev/null 2>&1
done
for e in c3_f32 c5_f32_end_to_end c2_f64_heavy_tail c2_f64_mw c2_f64_diag_noise c2_f64_factor_prior c2_f64_rowvecs c2_f64_dense_prior c4_f32 c4_f64_B4096 c4_f64_B2048 c4_f64_B1024 c3_f32_mw logpdf_only_c3_f32 c3_f32_B8 c5_shape_f32_B8 marginals_mean_c2_f64 marginals_var_c2_f64 \
         marginals_var_c2_f32 marginals_mean_c3_f32 marginals_var_c3_f32 marginals_var_D512_B16_f32 rand_c2_f64_S64 rand_c3_f32_S64 logpdf_grad_c2_f64 logpdf_grad_c2_f64_sweep_kernel logpdf_grad_c2_f32 logpdf_multi_c3_f32_S64 \
         update_factor_D128_k1_f64 update_factor_D128_k16_f64; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/sec_fetch_$e -- $B --secondary-only $e > $OUT/sec_fetch_$e.json 2>/dev/null
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/sec_write_$e -- $B --secondary-only $e > $OUT/sec_write_$e.json 2>/dev/null
done
# the headline kernels of this round: int8-sliced Gram (default) and the fp64 kernel (BLR_MI355X_NO_I8_GRAM=1)
BLR_MI355X_NO_I8_GRAM=1 python3 $R/bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | tail -1 > $OUT/bench_c2_f64_fp64kernel.json
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_c2_i8 -- $B --steps 5 --warmup 2 > /dev/null 2>&1
find $OUT -name "*agent_info*" -delete
ls -R $OUT | head -80
tail -1 $OUT/bench_c2_f64.out | cut -c1-600
