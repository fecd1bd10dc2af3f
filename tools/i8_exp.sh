#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/tools
{
for b in 16 64 256 1024 4096; do echo "=== B=$b"; timeout 120 ./i8_gram_st $b 4096 10 0 | grep "wave [0-7], mean\|int8 path" | cut -c1-140; done
for v in _e1 _e2 _e3; do echo "=== variant $v B=4096"; I8_SUSTAINED=1.5 timeout 120 ./i8_gram$v 4096 4096 4 0 | grep "wave [0-7], mean\|sustained\|int8 path" | cut -c1-200; 
 echo "=== variant $v B=16";  timeout 120 ./i8_gram$v 16 4096 4 0 | grep "wave [0-7], mean" | cut -c1-140; done
} 2>&1 | tee $R/gpurun_out/i8_exp.txt
