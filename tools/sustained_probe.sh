#!/bin/bash
# Runs on the GPU box (gpurun): the clock the part HOLDS under the headline kernel's load (MI355X_MICROARCH.md, DVFS give-back
# item 6: >= 2 s of back-to-back launches on random data, in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz,
# median over workgroups) -- the Gram ring loop alone (tools/ring_probe) and the whole fused kernel (tools/fused_bench, stamped
# diagnostic build + the plain build for the rate), two workgroups per CU, zero and random operands.
#   build here first:  cd tools && hipcc ... ring_probe.hip -o ring_probe -DRING_MWZ=true; ./b.sh fused_bench _st -DBLR_GRAM_STAMPS; ./b.sh fused_bench ""
#   gpurun --timeout 600 -- 'bash tools/sustained_probe.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/microbench
mkdir -p "$OUT"
cd $R/tools
S=${SUSTAIN:-2.5}
{
  echo "# tools/ring_probe 8192 1 $S   (random operands; every configuration $S s of back-to-back launches before the stamps are read)"
  timeout 200 ./ring_probe 8192 1 $S
  echo "# tools/ring_probe 8192 0 $S   (zero operands)"
  timeout 200 ./ring_probe 8192 0 $S
  echo "# tools/fused_bench_st 4096 4096 0 420   (whole fused kernel, stamped diagnostic build, random operands, 420 launches back to back)"
  timeout 200 ./fused_bench_st 4096 4096 0 420
  echo "# FB_ZERO=1 tools/fused_bench_st 4096 4096 0 420   (zero operands)"
  FB_ZERO=1 timeout 200 ./fused_bench_st 4096 4096 0 420
  echo "# tools/fused_bench 4096 4096 0 420   (the shipped kernel, no stamps: the rate the clock above belongs to)"
  timeout 200 ./fused_bench 4096 4096 0 420
  echo "# FB_ZERO=1 tools/fused_bench 4096 4096 0 420"
  FB_ZERO=1 timeout 200 ./fused_bench 4096 4096 0 420
} > $OUT/ring_probe_sustained.txt 2>&1
cat $OUT/ring_probe_sustained.txt
