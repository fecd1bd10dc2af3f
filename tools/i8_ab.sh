#!/bin/bash
# A/B of int8-kernel builds in one box: tools/i8_ab.sh <suffix> [<suffix> ...]  (binaries tools/i8_gram<suffix>, tools/i8_gram_st<suffix>)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/tools
mkdir -p $R/gpurun_out
{
for rep in 1 2; do
for s in "$@"; do
  [ "$s" = "-" ] && s=""
  echo "=== i8_gram$s (rep $rep)"
  timeout 120 ./i8_gram$s 4096 4096 20 0 | tail -2; 
done; done
for s in "$@"; do
  [ "$s" = "-" ] && s=""
  echo "=== i8_gram$s accuracy modes"
  for m in 2 4 1; do timeout 120 ./i8_gram$s 4096 4096 4 $m | tail -2; echo "rc=$?"; done
  timeout 120 ./i8_gram$s 512 4096 3 5 | tail -3; echo "rc=$?"
  timeout 120 ./i8_gram$s 256 16384 3 0 | tail -2; echo "rc=$?"
  timeout 120 ./i8_gram$s 512 1056 3 0 | tail -2; echo "rc=$?"
  I8_MW=1 timeout 120 ./i8_gram$s 1024 4096 4 0 | tail -2; echo "rc=$?"
  echo "=== i8_gram_st$s sustained"
  [ -x ./i8_gram_st$s ] && { I8_SUSTAINED=2.5 timeout 120 ./i8_gram_st$s 4096 4096 4 0 | grep "wave\|sustained" | cut -c1-330; I8_SUSTAINED=2.5 timeout 120 ./i8_gram_st$s 4096 4096 4 3 | grep "sustained"; }
done
} 2>&1 | tee $R/gpurun_out/i8_ab.txt
