#!/bin/bash
# development runs of the int8-sliced kernel in isolation (tools/i8_gram.hip): rates, agreement with the fp64 kernel, repair / hand-back paths
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/tools
mkdir -p $R/gpurun_out
{
for m in 0 2 4 1; do timeout 120 ./i8_gram 4096 4096 10 $m; echo "rc=$?"; done
timeout 120 ./i8_gram 512 4096 3 5; echo "rc=$?"
timeout 120 ./i8_gram 512 1024 3 0; echo "rc=$?"
timeout 120 ./i8_gram 256 16384 3 0; echo "rc=$?"
timeout 120 ./i8_gram 4096 4096 10 3; echo "rc=$?"
I8_MW=1 timeout 120 ./i8_gram 4096 4096 10 0; echo "rc=$?"
[ -x ./i8_gram_st ] && { I8_SUSTAINED=2.5 timeout 120 ./i8_gram_st 4096 4096 4 0 | grep "wave\|sustained"; I8_SUSTAINED=2.5 timeout 120 ./i8_gram_st 4096 4096 4 3 | grep "sustained"; }
} 2>&1 | tee $R/gpurun_out/i8_dev.txt
