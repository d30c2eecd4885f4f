#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/${1:-x6}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; echo "tests rc=$?" >> "$OUT/tests.log"
tail -3 "$OUT/tests.log"
TRACYHIP_HOST_TIMERS=1 python bench.py --workload decompose --decompose-steps 4 --cpu-sample 0 --extra-legs 0 > "$OUT/dec.json" 2> "$OUT/dec.err"
grep "^host" "$OUT/dec.err"
python bench.py --workload decompose --decompose-steps 4 --cpu-sample 0 > "$OUT/dec2.json" 2> "$OUT/dec2.err"
python bench.py --workload align --steps 10 --warmup 2 --cpu-sample 0 > "$OUT/al.json" 2> "$OUT/al.err"
