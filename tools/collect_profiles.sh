#!/bin/bash
# Runs on the GPU box (gpurun): bench line + rocprofv3 kernel stats + PMC passes for the headline workload (c2) and the
# two large-D shapes.  Everything lands in gpurun_out/profiles_raw/; tools/summarise_profiles.py turns it into profiles/.
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_raw
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_c2_f64.json 2> $OUT/bench_c2_f64.err
python3 $R/bench.py --dtype f32 --cpu-seconds 0 > $OUT/bench_c2_f32.json 2>/dev/null
python3 $R/bench.py --D 1024 --N 65536 --dtype f32 --noise diagonal --batch 1 --steps 20 --warmup 3 --cpu-seconds 0 > $OUT/bench_c3_f32.json 2>/dev/null
python3 $R/bench.py --D 2048 --N 16384 --dtype f32 --batch 1 --steps 20 --warmup 3 --cpu-seconds 0 > $OUT/bench_c5shape_f32.json 2>/dev/null
python3 $R/bench.py --D 64 --N 1024 --batch 8192 --cpu-seconds 0 > $OUT/bench_c4shape_f64.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $R/bench.py --cpu-seconds 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 $R/bench.py --D 1024 --N 65536 --dtype f32 --noise diagonal --batch 1 --steps 20 --warmup 3 --cpu-seconds 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python3 $R/bench.py --D 2048 --N 16384 --dtype f32 --batch 1 --steps 20 --warmup 3 --cpu-seconds 0 > /dev/null 2>&1
# PMC: separate passes (TCC FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c2 -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c2 -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq_c2 -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq_c3 -- python3 $R/bench.py --D 1024 --N 65536 --dtype f32 --noise diagonal --batch 1 --steps 5 --warmup 2 --cpu-seconds 0 > /dev/null 2>&1
find $OUT -name "*agent_info*" -delete
ls -R $OUT | head -60
cat $OUT/bench_c2_f64.json
