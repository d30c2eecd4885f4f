"""one line per run: decompose step, small batch and kernel-class timers (development tool; used with tools/ab.sh on one box)"""
import json
import subprocess
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "decompose", "--cpu-sample", "0", "--decompose-steps", "3"] + args, capture_output=True, text=True)
try:
    d = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
    sb = d.get("small_batch") or {}
    print("dec %.2f ms  small %s / %s  timers %s" % (d["ms_per_step"], sb.get("ms_per_step"), sb.get("ms_per_call_without_packing"), d["roofline"]["ms_per_step"]))
except Exception as e:  # noqa: BLE001
    print("failed:", e, r.stderr[-1500:])
