"""Where the GPU sits idle during a step: gaps between consecutive kernels / copies of a rocprofv3 trace.

  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -- python bench.py --workload decompose ...
  python tools/timeline_gaps.py /tmp/tr [anchor-kernel-substring [steps-back]]

Prints, for one step (from one occurrence of the anchor kernel to the next; steps-back = 0 is the last such interval), the
busy time, the idle time and the largest gaps with the activities on either side.
"""
import csv
import glob
import sys


def events(d):
    ev = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
    for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
    ev.sort()
    return ev


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "breakpoint_kernel"
    ev = events(d)
    idx = [i for i, e in enumerate(ev) if anchor in e[2]]
    if len(idx) < 2:
        print("anchor", anchor, "seen", len(idx), "times: need two")
        return
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if len(idx) < back + 2:
        print("anchor", anchor, "seen", len(idx), "times: need", back + 2)
        return
    a, b = idx[-2 - back], idx[-1 - back]
    step = ev[a:b]
    busy_end = step[0][1]
    busy = step[0][1] - step[0][0]
    gaps = []
    for k in range(1, len(step)):
        s, e, n = step[k]
        if s > busy_end:
            gaps.append((s - busy_end, step[k - 1][2], n))
            busy += e - s
        else:
            busy += max(0, e - busy_end)
        busy_end = max(busy_end, e)
    total = (busy_end - step[0][0]) / 1e6
    idle = sum(g[0] for g in gaps) / 1e6
    print("step %.2f ms: busy %.2f ms, idle %.2f ms in %d gaps (%d activities)" % (total, busy / 1e6, idle, len(gaps), len(step)))
    for g in sorted(gaps, reverse=True)[:25]:
        print("  %7.3f ms  after %-60s before %s" % (g[0] / 1e6, g[1][:60], g[2][:60]))
    print("in order of time (idle between two kernels, copies and fills skipped over):")
    last_kernel, acc = None, 0
    busy_end = step[0][1]
    for k in range(len(step)):
        s, e, n = step[k]
        if k and s > busy_end:
            acc += s - busy_end
        busy_end = max(busy_end, e)
        if not (n.startswith("COPY") or "rocclr" in n):
            if last_kernel is not None and acc > 100000:
                print("  %7.3f ms  between %-50s and %s" % (acc / 1e6, last_kernel[:50], n[:50]))
            last_kernel, acc = n, 0


if __name__ == "__main__":
    main()
