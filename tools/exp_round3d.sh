#!/bin/bash
# scratch: GPU tests + the two headline legs
set -u
OUT=/root/repo/gpurun_out/${1:-x3}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; echo "tests rc=$?" >> "$OUT/tests.log"
python bench.py --workload align --steps 10 --warmup 2 --cpu-sample 0 > "$OUT/al.json" 2> "$OUT/al.err"
python bench.py --workload decompose --decompose-steps 3 --cpu-sample 0 > "$OUT/dec.json" 2> "$OUT/dec.err"
tail -3 "$OUT/tests.log"
