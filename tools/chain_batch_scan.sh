#!/bin/bash
# Batched regressors at D > 128: shared factorisation launches (posterior_large_group) against one regressor at a time.
#   tools/chain_batch_scan.sh   (on the GPU box; prints ms per step = per batch)
cd "$(dirname "$0")/.."
run() {  # D N batch
  for g in 1 2 4; do
    printf "D=%d N=%d B=%d chain_batch=%d: " $1 $2 $3 $g
    BLR_MI355X_CHAIN_BATCH=$g python bench.py --D $1 --N $2 --dtype ${4:-f32} --noise diagonal --batch $3 --cpu-seconds 0 --secondary 0 --steps 20 --warmup 3 2>/dev/null \
      | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('%.3f ms/step, %.1f updates/s' % (d['ms_per_step'], d['value']))"
  done
}
run 1024 65536 4
run 2048 16384 4
run 512 16384 8
run 4096 8192 2
run 1024 16384 4 f64
