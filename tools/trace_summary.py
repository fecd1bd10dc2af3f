"""Summarise a rocprofv3 --kernel-trace CSV per posterior update (large-D path): python tools/trace_summary.py <dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "blr::" in r["Kernel_Name"]]
agg = collections.OrderedDict()
n = 0
for r in rows:
    k = r["Kernel_Name"].split("<")[0].replace("void blr::", "").split("(")[0]
    if k in ("prior_diag_kernel", "prior_copy_kernel"):
        n += 1
    agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
n = max(n, 1)
tot = 0.0
for k, v in agg.items():
    print("%-26s calls/update %5.1f  us/update %8.1f  avg us %7.1f" % (k, len(v) / n, sum(v) / n, sum(v) / len(v)))
    tot += sum(v) / n
print("updates", n, "kernel us/update", round(tot, 1))
