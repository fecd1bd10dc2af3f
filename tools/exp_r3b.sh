#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}/tools
OUT=../gpurun_out/exp_r3b; mkdir -p $OUT
{ echo "== chol lean"; timeout 120 ./chol_bench_st_lean
  echo "== c2 f64 tout (old chol)"; timeout 120 ./fused_bench_tout 4096 4096 0 10 | tail -1
  echo "== c2 f64 lean"; timeout 120 ./fused_bench_lean 4096 4096 0 10 | tail -1
  echo "== c2 f64 lean diag-noise"; timeout 120 ./fused_bench_lean 4096 4096 1 10 | tail -1
  echo "== c2 f32 lean"; timeout 120 ./fused_bench_f32_lean 4096 4096 0 10 | tail -1
  echo "== stamps lean"; timeout 120 ./fused_bench_stamps_lean 4096 4096 0 3 | tail -7
} > $OUT/out.txt 2>&1
cat $OUT/out.txt
