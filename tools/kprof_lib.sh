#!/bin/bash
# tools/kprof_lib.sh <lib.so relative to repo | -> <entry>: top kernels of one secondary entry under a given library
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "$1" != "-" ]; then export BLR_MI355X_LIB=$R/$1; fi
e=$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp_$e
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_$e -- python3 $R/bench.py --secondary-only $e > /tmp/kp_$e.log 2>&1
f=$(find /tmp/kp_$e -name "*kernel_stats.csv" | head -1)
echo "== $1 $e"
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:3]:
    print("  %-70s calls %5s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
