#!/bin/bash
# Round profiles (run on the GPU box through gpurun): rocprofv3 kernel stats and PMC passes of bench.py's three workloads.
#   align (configs[1], the headline): headline leg only (--certificate-leg 0 --lanes-leg 0) so that per-kernel averages and counters
#     are not blended with the other legs' launches of the same kernels; one more kernel-trace pass of the DEFAULT align legs is kept
#     and summarised per (kernel, grid size)
#   decompose (configs[2], 100 000 traces) and all-pairs (configs[4], 1000 traces): kernel stats + the same counters
# Counters: separate --pmc passes with --kernel-trace only (WRITE_SIZE, FETCH_SIZE; SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES).
# Outputs under gpurun_out/prof_round/; tools/profile_summarise.py turns them into the files kept in profiles/.
set -u
TAG=${1:-r06}
OUT=/root/repo/gpurun_out/prof_round
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=/root/repo/bench.py
AL="--workload align --steps 3 --warmup 1 --certificate-leg 0 --lanes-leg 0"
DE="--workload decompose --decompose-steps 2 --extra-legs 0"
DEX="--workload decompose --decompose-steps 3"  # the line kept for the record: with the certificate / two-lane / small-batch legs
AP="--workload allpairs --allpairs-steps 2"
SE="--workload seedextend --seedextend-steps 1 --seedextend-traces 31250"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python $B $AL > "$OUT/bench_line.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dec_stats" -- python $B $DE > "$OUT/dec_line_under_rocprof.json" 2> "$OUT/dec.err"
python $B $DEX > "$OUT/dec_line.json" 2> "$OUT/dec_plain.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ap_stats" -- python $B $AP > "$OUT/ap_line.json" 2> "$OUT/ap.err"
for w in bench dec ap se; do
  case $w in bench) ARGS="$AL --steps 2 --cpu-sample 0";; dec) ARGS="$DE --cpu-sample 0";; ap) ARGS="$AP --cpu-sample 0";; se) ARGS="$SE --cpu-sample 0";; esac
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${w}_$c" -- python $B $ARGS > /dev/null 2> "$OUT/pmc_${w}_$c.err"
  done
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_${w}_valu" -- python $B $ARGS > /dev/null 2> "$OUT/pmc_${w}_valu.err"
done
# stall counters (where the wave cycles go: parked in s_waitcnt / s_barrier, issue stalls, LDS) and effective clocks, align and decompose
for w in bench dec; do
  case $w in bench) ARGS="$AL --steps 2 --cpu-sample 0";; dec) ARGS="$DE --cpu-sample 0";; esac
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU \
    --output-format csv -d "$OUT/pmc_${w}_stallA" -- python $B $ARGS > /dev/null 2> "$OUT/pmc_${w}_stallA.err"
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d "$OUT/pmc_${w}_stallB" -- python $B $ARGS > /dev/null 2> "$OUT/pmc_${w}_stallB.err"
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_${w}_clock" -- python $B $ARGS > /dev/null 2> "$OUT/pmc_${w}_clock.err"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_default_stats" -- python $B --workload align --steps 3 --warmup 1 > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"
python $B --full-line "$OUT/bench_plain_full.json" > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
# where the GPU waits for the host: kernel + memory-copy timelines of one step of each pipeline (tools/timeline_gaps.py)
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/tl_dec" -- python $B $DE --cpu-sample 0 > /dev/null 2> "$OUT/tl_dec.err"
python /root/repo/tools/timeline_gaps.py "$OUT/tl_dec" > "$OUT/decompose_timeline_gaps.txt" 2>&1
python /root/repo/tools/timeline_dump.py "$OUT/tl_dec" > "$OUT/decompose_timeline.txt" 2>&1
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
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/tl_al" -- python $B $AL --cpu-sample 0 > /dev/null 2> "$OUT/tl_al.err"
python /root/repo/tools/timeline_gaps.py "$OUT/tl_al" encode_codes_kernel > "$OUT/align_timeline_gaps.txt" 2>&1
python /root/repo/tools/timeline_dump.py "$OUT/tl_al" > "$OUT/align_timeline.txt" 2>&1
# the shard one of 8 GPUs gets from the 100 000-trace decompose job, as a job of its own: what a step costs once the kernels are short
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/tl_small" -- python $B --workload decompose --decompose-traces 12500 --decompose-steps 3 --extra-legs 0 --cpu-sample 0 > /dev/null 2> "$OUT/tl_small.err"
python /root/repo/tools/timeline_gaps.py "$OUT/tl_small" > "$OUT/decompose_small_batch_timeline_gaps.txt" 2>&1
python /root/repo/tools/timeline_dump.py "$OUT/tl_small" > "$OUT/decompose_small_batch_timeline.txt" 2>&1
rm -rf "$OUT/tl_dec" "$OUT/tl_al" "$OUT/tl_small"
find "$OUT" -name "*.csv" -size +40M -delete
ls "$OUT" | head -60
