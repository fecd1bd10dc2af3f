import csv, sys, glob, collections
d = sys.argv[1]; kern = sys.argv[2]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
