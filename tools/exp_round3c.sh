#!/bin/bash
# scratch: timelines of the decompose and align steps (where the GPU waits for the host)
set -u
OUT=/root/repo/gpurun_out/x2
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TRACYHIP_HOST_TIMERS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/dec" -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 > "$OUT/dec.json" 2> "$OUT/dec.err"
python /root/repo/tools/timeline_gaps.py "$OUT/dec" > "$OUT/dec_gaps.txt" 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/al" -- python /root/repo/bench.py --workload align --steps 3 --warmup 1 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0 > "$OUT/al.json" 2> "$OUT/al.err"
python /root/repo/tools/timeline_gaps.py "$OUT/al" encode_codes_kernel > "$OUT/al_gaps.txt" 2>&1
find "$OUT" -name "*.csv" -size +30M -delete
