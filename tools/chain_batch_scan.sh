#!/bin/bash
# Batched regressors at D > 128: shared factorisation launches (posterior_large_group) against one regressor at a time.
#   tools/chain_batch_scan.sh   (on the GPU box; prints ms per step = per batch)
cd "$(dirname "$0")/.."
run() {  # D N batch dtype groups...
  D=$1; N=$2; B=$3; dt=$4; shift 4
  for g in "$@"; do
    printf "D=%d N=%d B=%d %s chain_batch=%d: " $D $N $B $dt $g
    BLR_MI355X_CHAIN_BATCH=$g python bench.py --D $D --N $N --dtype $dt --noise diagonal --batch $B --cpu-seconds 0 --secondary 0 --steps 20 --warmup 3 2>/dev/null \
      | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('%.3f ms/step, %.1f updates/s' % (d['ms_per_step'], d['value']))"
  done
}
run 256 4096 128 f32 8 16 32 64 128
run 384 4096 128 f32 16 32 64
run 512 16384 64 f32 16 32 64
run 1024 16384 32 f32 8 16 32
run 256 4096 128 f64 16 32 64
