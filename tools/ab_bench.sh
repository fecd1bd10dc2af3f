#!/bin/bash
# same-box A/B of the driver's headline line with two libraries:  tools/ab_bench.sh <libA.so> <libB.so>   (paths relative to the repo root; "-" = the built one)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset BLR_MI355X_LIB; else export BLR_MI355X_LIB=$R/$lib; fi
    python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$lib rep $rep: value %.0f  ms_per_step %.4f  kernel median %.4f min %.4f  hbm_frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_median'], r['kernel_ms_min'], r['hbm_frac']))"
  done
done 2>&1 | tee $R/gpurun_out/ab_bench.txt
