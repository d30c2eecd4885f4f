"""One line per bench.py invocation for tools/ab.sh (development tool): python tools/ab_line.py <bench args>"""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py"] + sys.argv[1:], capture_output=True, text=True).stdout
d = json.loads([l for l in out.split("\n") if l.startswith("{")][-1])
sb = d.get("small_batch") or {}
pl = d.get("pipeline") or {}
print("ms_per_step %.3f  frac %.3f  kernel_ms %.3f  small_batch %s  fallback %s loser_won %s" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("avg_launch_ms", 0.0), sb.get("ms_per_step"),
      pl.get("traces_to_host_planned_tiers", pl.get("fallback_traces")), {k: v for k, v in pl.items() if "vote" in k or "loser" in k or "pruned" in k}))
