#!/bin/bash
# Gram launch geometry variants (f32): k-steps per stage x workgroups per CU
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/exp_r3c; mkdir -p $OUT
for v in base ks4w3 ks4w4 ks8w3; do
  if [ $v = base ]; then unset BLR_MI355X_LIB; else export BLR_MI355X_LIB=$PWD/bayesianlinearregressors.jl_amd/csrc/exp/libblr_$v.so; fi
  for c in c3 c5; do
    echo "== $v $c"; python bench.py --config $c --steps 20 --warmup 3 --cpu-seconds 0 --secondary 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
  done
done > $OUT/out.txt 2>&1
cat $OUT/out.txt
