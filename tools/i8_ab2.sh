#!/bin/bash
# sustained A/B of stamped int8-kernel builds in one box: tools/i8_ab2.sh <suffix> ...   (binaries tools/i8_gram_st<suffix>)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/tools
{
for rep in 1 2; do
for s in "$@"; do
  [ "$s" = "-" ] && s=""
  echo "=== i8_gram_st$s (rep $rep)"
  I8_SUSTAINED=${SUST:-2.0} timeout 120 ./i8_gram_st$s 4096 4096 4 0 | grep "wave [0-7], mean\|sustained\|int8 vs" | cut -c1-250
done; done
} 2>&1 | tee $R/gpurun_out/i8_ab2.txt
