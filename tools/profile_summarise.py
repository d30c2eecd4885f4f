#!/usr/bin/env python
"""Turn gpurun_out/prof_round (tools/profile_round.sh) into the summaries committed under profiles/:
   rNN_{bench,decompose,allpairs}_kernel_stats.csv  -- rocprofv3 --kernel-trace --stats of each workload
   rNN_pmc_hbm[_decompose|_allpairs].json   -- WRITE_SIZE / FETCH_SIZE per kernel, per launch (separate PMC passes; KB -> bytes)
   rNN_pmc_valu[_decompose|_allpairs].json  -- VALU instructions / busy cycles per kernel, per launch
   rNN_*_line_under_rocprof.json (the JSON lines those runs printed), rNN_bench_line.json (the default command, no profiler)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_round")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"


def one(pattern):
    """the NEWEST match: gpurun merges a run's files into gpurun_out/, where older rounds' files (other process ids) may still lie"""
    f = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)
    return f[-1] if f else None


def last_json_line(path):
    try:  # (a whole file of indented JSON: bench.py --full-line)
        return json.load(open(path))
    except ValueError:
        pass
    for ln in reversed(open(path).read().strip().split("\n")):
        if ln.startswith("{"):
            return json.loads(ln)
    return None


for name, sub in (("bench", "bench_stats"), ("decompose", "dec_stats"), ("allpairs", "ap_stats")):
    f = one(sub + "/*/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(DST, "%s_%s_kernel_stats.csv" % (tag, name)))
for src, dst in (("bench_line.json", "bench_line_under_rocprof"), ("bench_plain.json", "bench_line"), ("bench_plain_full.json", "bench_line_full"), ("dec_line_under_rocprof.json", "decompose_line_under_rocprof"), ("dec_line.json", "decompose_line"),
                 ("ap_line.json", "allpairs_line_under_rocprof")):
    p = os.path.join(SRC, src)
    if os.path.exists(p):
        d = last_json_line(p)
        if d:
            json.dump(d, open(os.path.join(DST, "%s_%s.json" % (tag, dst)), "w"), indent=1)


def per_kernel(pass_dir, scale):
    f = one(pass_dir + "/*/*counter_collection.csv")
    if not f:
        return []
    acc = collections.OrderedDict()
    # one row per (kernel, grid size, counter): the full sweeps and the prefix launch of one instantiation differ in their grids
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"], r.get("Grid_Size", ""), r["Counter_Name"])
        a = acc.setdefault(k, {"sum": 0.0, "disp": set()})
        a["sum"] += float(r["Counter_Value"])
        a["disp"].add(r["Dispatch_Id"])
        dur[(r["Kernel_Name"], r.get("Grid_Size", ""))][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out = []
    for (kern, grid, ctr), a in acc.items():
        if not kern.startswith("void tracyhip") and "tracyhip" not in kern and "anonymous" not in kern:
            continue
        n = max(len(a["disp"]), 1)
        ms = list(dur[(kern, grid)].values())
        out.append({"counter": ctr, "kernel": kern, "grid": int(grid) if grid else None, "launches": n, "per_launch": a["sum"] / n * scale,
                    "avg_launch_ms": sum(ms) / len(ms) if ms else None})
    return out


# the default command (both legs): per (kernel, grid) launch statistics from the kernel trace
f = one("bench_default_stats/*/*kernel_trace.csv")
if f:
    grp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "tracyhip" not in r["Kernel_Name"]:
            continue
        grp.setdefault((r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", ""))), []).append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    with open(os.path.join(DST, "%s_bench_default_kernel_by_grid.csv" % tag), "w") as o:
        o.write("kernel,grid_size,launches,avg_ms,min_ms,max_ms\n")
        for (k, g), v in grp.items():
            o.write('"%s",%s,%d,%.4f,%.4f,%.4f\n' % (k, g, len(v), sum(v) / len(v), min(v), max(v)))
    p = os.path.join(SRC, "bench_default_line.json")
    if os.path.exists(p) and last_json_line(p):
        json.dump(last_json_line(p), open(os.path.join(DST, "%s_bench_default_line_under_rocprof.json" % tag), "w"), indent=1)

for w, suffix in (("bench", ""), ("dec", "_decompose"), ("ap", "_allpairs"), ("se", "_seedextend")):
    hbm = []
    for c in ("WRITE_SIZE", "FETCH_SIZE"):
        for r in per_kernel("pmc_%s_%s" % (w, c), 1024.0):  # the counters report KB
            r["bytes"] = r.pop("per_launch")
            hbm.append(r)
    if hbm:
        json.dump(hbm, open(os.path.join(DST, "%s_pmc_hbm%s.json" % (tag, suffix)), "w"), indent=1)
    valu = per_kernel("pmc_%s_valu" % w, 1.0)
    if valu:
        json.dump(valu, open(os.path.join(DST, "%s_pmc_valu%s.json" % (tag, suffix)), "w"), indent=1)
# stall counters (two SQ passes) and effective clocks (GRBM_GUI_ACTIVE over the XCDs / duration) per kernel and grid
for w, suffix in (("bench", ""), ("dec", "_decompose")):
    rows = []
    for ps in ("stallA", "stallB", "clock"):
        rows += per_kernel("pmc_%s_%s" % (w, ps), 1.0)
    if rows:
        for r in rows:
            if r["counter"] == "GRBM_GUI_ACTIVE" and r["avg_launch_ms"]:
                r["effective_clock_ghz"] = round(r["per_launch"] / 8.0 / (r["avg_launch_ms"] * 1e6), 3)  # (summed over the eight XCDs)
        json.dump(rows, open(os.path.join(DST, "%s_pmc_stalls%s.json" % (tag, suffix)), "w"), indent=1)
for nm in ("decompose_timeline.txt", "align_timeline.txt", "decompose_small_batch_timeline.txt"):
    if os.path.exists(os.path.join(SRC, nm)):
        shutil.copy(os.path.join(SRC, nm), os.path.join(DST, "%s_%s" % (tag, nm)))
for nm in ("decompose_timeline_gaps.txt", "align_timeline_gaps.txt", "decompose_small_batch_timeline_gaps.txt"):
    if os.path.exists(os.path.join(SRC, nm)):
        shutil.copy(os.path.join(SRC, nm), os.path.join(DST, "%s_%s" % (tag, nm)))
print("profiles written for", tag, ":", sorted(f for f in os.listdir(DST) if f.startswith(tag)))
