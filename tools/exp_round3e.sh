#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/${1:-x5}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /root/repo
TRACYHIP_HOST_TIMERS=1 python bench.py --workload decompose --decompose-steps 3 --cpu-sample 0 --extra-legs 0 > "$OUT/dec.json" 2> "$OUT/dec.err"
grep "^host" "$OUT/dec.err"
