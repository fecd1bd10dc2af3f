#!/bin/bash
# Kernel timeline of one step of a bench workload (GPU box):  bash tools/timeline.sh c3 [extra bench args]
# rocprofv3 --kernel-trace start/end stamps -> gpurun_out/timeline_<cfg>.txt (the last step: name, queue, start, duration in us)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c3}; shift || true
OUT=$R/gpurun_out/timeline_raw_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --cpu-seconds 0 --secondary 0 --config $CFG --steps 3 --warmup 2 "$@" > $OUT/bench.log 2>&1
python3 $R/tools/timeline.py $OUT > $R/gpurun_out/timeline_$CFG.txt
tail -n 120 $R/gpurun_out/timeline_$CFG.txt
