"""One step of a rocprofv3 kernel trace as a table: start / end offsets (ms) from the anchor kernel, stream (queue), kernel, grid.

  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python bench.py --workload decompose ...
  python tools/timeline_dump.py /tmp/tr [anchor-kernel-substring [steps-back]]
"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "encode_codes_kernel"
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    ev = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", ""),
                       r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", "")))
    ev.sort()
    idx = [i for i, e in enumerate(ev) if anchor in e[2]]
    if len(idx) < back + 2:
        print("anchor seen", len(idx), "times")
        return
    a, b = idx[-2 - back], idx[-1 - back]
    t0 = ev[a][0]
    print("step of %.2f ms, %d kernels" % ((ev[b][0] - t0) / 1e6, b - a))
    for s, e, n, q, st, g, lds, vg in ev[a:b]:
        short = n.replace("(anonymous namespace)::", "").replace("tracyhip::", "").replace("void ", "")
        short = short.split("(")[0][:58]
        print("%8.3f %8.3f %7.3f  q%-3s s%-3s grid %-9s lds %-6s vgpr %-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, st, g, lds, vg, short))


main()
