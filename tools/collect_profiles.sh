#!/bin/bash
# Runs on the GPU box (gpurun): bench lines + rocprofv3 kernel stats + PMC passes for the headline workload (c2) and the
# secondary shapes.  Everything lands in gpurun_out/profiles_raw/; tools/summarise_profiles.py turns it into profiles/.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_raw
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-seconds 0 --secondary 0"
python3 $R/bench.py --cpu-seconds 12 > $OUT/bench_c2_f64.json 2> $OUT/bench_c2_f64.err
$B --dtype f32 > $OUT/bench_c2_f32.json 2>/dev/null
$B --config c3 --steps 20 --warmup 3 > $OUT/bench_c3_f32.json 2>/dev/null
$B --config c5 --steps 20 --warmup 3 > $OUT/bench_c5_f32.json 2>/dev/null
$B --config c4 > $OUT/bench_c4_f64.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- $B --config c3 --steps 20 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- $B --config c5 --steps 20 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c4 -- $B --config c4 > /dev/null 2>&1
# PMC: separate passes (TCC FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only
SQ="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c2 -- $B --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c2 -- $B --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_c2 -- $B --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_c2f32 -- $B --dtype f32 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_c3 -- $B --config c3 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq_c4 -- $B --config c4 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c4 -- $B --config c4 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c4 -- $B --config c4 --steps 5 --warmup 2 > /dev/null 2>&1
find $OUT -name "*agent_info*" -delete
ls -R $OUT | head -80
cat $OUT/bench_c2_f64.json | cut -c1-400
