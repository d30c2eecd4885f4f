#!/bin/bash
set -u
TAG=${1:-r06}
OUT=/root/repo/gpurun_out/prof_round
cd /tmp && export TMPDIR=/tmp
B=/root/repo/bench.py
AL="--workload align --steps 3 --warmup 1 --certificate-leg 0 --lanes-leg 0"
DE="--workload decompose --decompose-steps 2 --extra-legs 0"
DEX="--workload decompose --decompose-steps 3"  # the line kept for the record: with the certificate / two-lane / small-batch legs
AP="--workload allpairs --allpairs-steps 2"
SE="--workload seedextend --seedextend-steps 1 --seedextend-traces 31250"
mkdir -p "$OUT"
# where the GPU waits for the host: kernel + memory-copy timelines of one step of each pipeline (tools/timeline_gaps.py)
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/tl_dec" -- python $B $DE --cpu-sample 0 > /dev/null 2> "$OUT/tl_dec.err"
# (steps-back 1: the last timed step, from its first kernel to the first kernel of the step after it -- the run ends with one more,
# untimed step whose gather is checked on the host: that one is not a step's timeline)
python /root/repo/tools/timeline_gaps.py "$OUT/tl_dec" encode_codes_kernel 1 > "$OUT/decompose_timeline_gaps.txt" 2>&1
python /root/repo/tools/timeline_dump.py "$OUT/tl_dec" encode_codes_kernel 1 > "$OUT/decompose_timeline.txt" 2>&1
# host synchronisations of a call: counted by the library (tracyhip_last_call_stats), as the bench line of the same workload printed them
python - "$OUT/dec_line.json" >> "$OUT/decompose_timeline_gaps.txt" <<'PYEOF'
import json, sys
try:
    d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    p = d["pipeline"]
    print("host synchronisations per tracyhip_decompose_traces call (tracyhip_last_call_stats): %d; stream-ordered: %d; traces handed to the host-planned tiers: %d; small batch (%d traces): %.2f ms per step = %.2f x an eighth of the full step"
          % (p["host_syncs_per_call"], p["stream_ordered"], p["traces_to_host_planned_tiers"], d.get("small_batch", {}).get("traces", 0),
             d.get("small_batch", {}).get("ms_per_step", 0.0), d.get("small_batch", {}).get("vs_eighth_of_the_full_step", 0.0)))
except Exception as e:  # noqa: BLE001
    print("(no pipeline object in the bench line: %s)" % e)
PYEOF
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/tl_al" -- python $B $AL --cpu-sample 0 --alone-steps 0 > /dev/null 2> "$OUT/tl_al.err"
python /root/repo/tools/timeline_gaps.py "$OUT/tl_al" encode_codes_kernel 1 > "$OUT/align_timeline_gaps.txt" 2>&1
python /root/repo/tools/timeline_dump.py "$OUT/tl_al" encode_codes_kernel 1 > "$OUT/align_timeline.txt" 2>&1
# the shard one of 8 GPUs gets from the 100 000-trace decompose job, as a job of its own: what a step costs once the kernels are short
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/tl_small" -- python $B --workload decompose --decompose-traces 12500 --decompose-steps 3 --extra-legs 0 --cpu-sample 0 > /dev/null 2> "$OUT/tl_small.err"
python /root/repo/tools/timeline_gaps.py "$OUT/tl_small" encode_codes_kernel 1 > "$OUT/decompose_small_batch_timeline_gaps.txt" 2>&1
python /root/repo/tools/timeline_dump.py "$OUT/tl_small" encode_codes_kernel 1 > "$OUT/decompose_small_batch_timeline.txt" 2>&1
rm -rf "$OUT/tl_dec" "$OUT/tl_al" "$OUT/tl_small"
