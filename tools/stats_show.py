"""Print the top rows of a rocprofv3 --stats kernel_stats.csv: name, calls, average us, us per bench call (argv[2] = calls)."""
import csv, sys
calls = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in list(csv.DictReader(open(sys.argv[1])))[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"]) / 1e3), "%9.1f us/call" % (float(r["TotalDurationNs"]) / 1e3 / calls))
