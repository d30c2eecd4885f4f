"""per-(kernel, grid size) sums of the counters of a rocprofv3 --pmc pass (counter_collection.csv): kernel name, grid, dispatches,
counter averages per dispatch, average duration -- the full sweeps and the prefix launch of one kernel instantiation are separate rows.

  python tools/pmc_by_kernel.py <pass dir> [kernel-name substring] [--json]"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
as_json = "--json" in sys.argv
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
n = collections.Counter()
seen = set()
for row in csv.DictReader(open(f[0])):
    k = (row["Kernel_Name"], row.get("Grid_Size", ""))
    if pat and pat not in k[0]:
        continue
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (k, row["Dispatch_Id"])
    if key not in seen:
        seen.add(key)
        n[k] += 1
        dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
out = []
for k in sorted(acc):
    rec = {"kernel": k[0], "grid": k[1], "dispatches": n[k], "avg_ms": round(dur[k] / n[k], 4), "per_dispatch": {c: round(v / n[k]) for c, v in acc[k].items()}}
    out.append(rec)
    if not as_json:
        print(k[0][:100], "grid", k[1], "dispatches", n[k], "avg_ms", rec["avg_ms"], rec["per_dispatch"])
if as_json:
    print(json.dumps(out, indent=1))
