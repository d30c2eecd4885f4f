"""Gaps and durations around the decompose_kernel launches of a rocprofv3 --kernel-trace --memory-copy-trace run.

  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -- python bench.py --workload decompose ...
  python tools/timeline_gaps.py /tmp/tr
"""
import csv, sys, glob
for d in sys.argv[1:]:
    ks = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    ms = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
    ev = []
    for f in ks:
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
    for f in ms:
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") ))
    ev.sort()
    idx = [i for i, e in enumerate(ev) if "decompose_kernel" in e[2]]
    print(d, len(ev), "events; decompose launches", len(idx))
    for i in idx[-2:]:
        for k in range(max(0, i - 6), min(len(ev), i + 12)):
            s, e, n = ev[k]
            gap = (s - ev[k - 1][1]) / 1e6 if k else 0
            print("   gap %8.3f ms  dur %8.3f ms  %s" % (gap, (e - s) / 1e6, n))
        print("   ---")
