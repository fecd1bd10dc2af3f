#!/bin/bash
# Runs on the GPU box (gpurun): the f64 / f32 matrix- and vector-pipe microbenchmarks whose numbers DESIGN.md and bench.py
# quote.  stdout of each lands in gpurun_out/microbench/; tools/summarise_profiles.py copies them to profiles/.
#   gpurun --timeout 900 -- 'bash tools/run_microbench.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/microbench
mkdir -p "$OUT"
cd $R/tools
for t in mfma_peak mfma_valu_mix dpp_fmac64 dpp_fmac32 mfma_f64_probe launch_probe gridbar_probe; do
  [ -x ./$t ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $t.hip -o $t 2>/dev/null
  timeout 300 ./$t > $OUT/$t.txt 2>&1
  echo "== $t rc=$?"
done
# the panel step of the large-D Cholesky in isolation (tools/pb.sh builds both): timing, then section sums + time line
[ -x ./panel_bench ] || ./pb.sh > /dev/null 2>&1
timeout 120 ./panel_bench 120 > $OUT/panel_bench.txt 2>&1
timeout 120 ./panel_bench_st 120 > $OUT/panel_bench_stamps.txt 2>&1
echo "== panel_bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_mfma_peak -- $R/tools/mfma_peak 2000 > $OUT/mfma_peak_traced.txt 2>&1
find $OUT -name "*agent_info*" -delete
cat $OUT/mfma_f64_probe.txt
