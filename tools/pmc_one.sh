#!/bin/bash
# SQ counters of one kernel in one secondary entry: tools/pmc_one.sh <entry> <kernel substring>
R=${GRAFT_REPO_ROOT:-$(pwd)}
e=$1; k=$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc1 -- python3 $R/bench.py --secondary-only $e > /tmp/pmc1.log 2>&1
f=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1)
python3 - "$f" "$k" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
acc=collections.defaultdict(float); n=collections.Counter()
for r in rows:
    acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for c in acc: print(c, acc[c]/n[c], "over", n[c], "dispatches")
PY
