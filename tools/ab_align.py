"""one line per run: align step, sweep interval and the sweeps alone (development tool; used with tools/ab.sh on one box)"""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10000").split(",")]:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "align", "--traces", str(n), "--steps", "6", "--warmup", "2", "--certificate-leg", "0",
                        "--lanes-leg", "0", "--cpu-sample", "0"] + sys.argv[2:], capture_output=True, text=True)
    try:
        d = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
        q = d["roofline"]
        print("traces %6d step %.3f ms  sweeps %.3f (frac %.3f)  alone %.3f (frac %.3f)" % (n, d["ms_per_step"], q["ms_score"], q["frac"], q["dominant_kernel_alone_ms"], q["dominant_kernel_alone_frac"]))
    except Exception as e:  # noqa: BLE001
        print("failed:", e, r.stderr[-800:])
