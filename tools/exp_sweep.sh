#!/bin/bash
# development experiment (run through gpurun): rebuild the library with -DTRACY_EXP=n variants of the 16-bit sweep and time
# the headline leg; variants other than 0 compute wrong results on purpose (they leave work out to see what it costs)
cd /root/repo
for e in "$@"; do
  TRACYHIP_CXXFLAGS="-DTRACY_EXP=$e" python tracy_amd/build.py --force > /dev/null 2>&1
  python - <<PY
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--workload", "align", "--certificate-leg", "0", "--lanes-leg", "0", "--cpu-sample", "0", "--steps", "5"], capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().split("\n")[-1])
    print("EXP $e: score %.2f ms  step %.2f ms" % (d["roofline"]["ms_per_step"]["score"], d["ms_per_step"]))
except Exception as ex:
    print("EXP $e failed", ex, out.stderr[-300:])
PY
done
python tracy_amd/build.py --force > /dev/null 2>&1
