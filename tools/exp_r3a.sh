#!/bin/bash
# round-3 experiment batch A (GPU box): c2 fused kernel variants + chol bench
cd ${GRAFT_REPO_ROOT:-/root/repo}/tools
OUT=../gpurun_out/exp_r3a; mkdir -p $OUT
for v in base tout la la_tout la_tout_p1 la_tout_p3 tout_p3; do
  echo "== $v"; timeout 120 ./fused_bench_$v 4096 4096 0 10 2>&1 | tail -3
done > $OUT/c2_f64.txt 2>&1
echo "== la_tout grid512" >> $OUT/c2_f64.txt; FB_GRID=512 timeout 120 ./fused_bench_la_tout 4096 4096 0 10 >> $OUT/c2_f64.txt 2>&1
echo "== la_tout_p3 grid512" >> $OUT/c2_f64.txt; FB_GRID=512 timeout 120 ./fused_bench_la_tout_p3 4096 4096 0 10 >> $OUT/c2_f64.txt 2>&1
for v in base la_tout la_tout_p3; do echo "== f32 $v"; timeout 120 ./fused_bench_f32_$v 4096 4096 0 10 2>&1 | tail -2; done > $OUT/c2_f32.txt 2>&1
for v in st_la0 st_la1; do echo "== $v"; timeout 120 ./chol_bench_$v 2>&1; done > $OUT/chol.txt 2>&1
for v in stamps_base stamps_la_tout_p3; do echo "== $v"; timeout 120 ./fused_bench_$v 4096 4096 0 3 2>&1 | tail -8; done > $OUT/c2_stamps.txt 2>&1
cat $OUT/c2_f64.txt $OUT/c2_f32.txt $OUT/chol.txt $OUT/c2_stamps.txt
