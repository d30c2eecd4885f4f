"""per-kernel sums of the counters of a rocprofv3 --pmc pass (counter_collection.csv): kernel name, dispatches, counter sums"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]
    if pat and pat not in k:
        continue
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (k, row["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        n[k] += 1
for k in sorted(acc):
    print(k[:100], "dispatches", n[k], {c: round(v / n[k]) for c, v in acc[k].items()})
