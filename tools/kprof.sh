#!/bin/bash
# kernel-trace summary of one secondary bench entry:  tools/kprof.sh <entry> [more bench args]   -> gpurun_out/kprof_<entry>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
e=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kprof_$e
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof_$e -- python3 $R/bench.py --secondary-only $e "$@" > /tmp/kprof_$e.log 2>&1
f=$(find /tmp/kprof_$e -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out
python3 - "$f" > $R/gpurun_out/kprof_$e.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-90s calls %5s avg %8.1f min %8.1f max %8.1f us  total %6.2f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["Percentage"])))
PY
cat $R/gpurun_out/kprof_$e.txt
