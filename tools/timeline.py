"""Prints the kernel timeline of the LAST step found in a rocprofv3 --kernel-trace csv directory (tools/timeline.sh)."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("<")[0].split("(")[0],
                         r.get("Queue_Id", "?"), r.get("Workgroup_Size_X", "?"), r.get("Grid_Size_X", "?")))
rows.sort()
# a step starts at the last colstats_kernel
starts = [i for i, r in enumerate(rows) if "colstats" in r[2]]
if not starts:
    starts = [max(0, len(rows) - 150)]
i0 = starts[-1]
t0 = rows[i0][0]
print(f"{'kernel':42s} {'queue':>6s} {'grid':>9s} {'start_us':>9s} {'dur_us':>8s} {'end_us':>9s}")
for s, e, n, q, wg, grid in rows[i0:]:
    print(f"{n[-42:]:42s} {q:>6s} {grid:>9s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(e - t0) / 1e3:9.1f}")
