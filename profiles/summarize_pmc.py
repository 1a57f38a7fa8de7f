"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per gatmi kernel."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gatmi" not in k:
            continue
        acc[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("    %-22s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
