#!/bin/bash
# kernel-time summary of one bench configuration (GPU box): tools/kstats.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
rm -rf $R/gpurun_out/st_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_$tag -- python3 $R/bench.py "$@" --cpu-seconds 0 --secondary 0 > /dev/null 2>&1
f=$(find $R/gpurun_out/st_$tag -name "*kernel_stats.csv" | head -1)
echo "== $tag"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(r["Name"][:58].ljust(58), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), "us", r["Percentage"])
PY
