#!/bin/bash
# Round profiles (run on the GPU box through gpurun): rocprofv3 kernel stats and PMC passes of bench.py's three workloads.
#   align (configs[1], the headline): headline leg only (--certificate-leg 0 --lanes-leg 0) so that per-kernel averages and counters
#     are not blended with the other legs' launches of the same kernels; one more kernel-trace pass of the DEFAULT align legs is kept
#     and summarised per (kernel, grid size)
#   decompose (configs[2], 100 000 traces) and all-pairs (configs[4], 1000 traces): kernel stats + the same counters
# Counters: separate --pmc passes with --kernel-trace only (WRITE_SIZE, FETCH_SIZE; SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES).
# Outputs under gpurun_out/prof_round/; tools/profile_summarise.py turns them into the files kept in profiles/.
set -u
TAG=${1:-r03}
OUT=/root/repo/gpurun_out/prof_round
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=/root/repo/bench.py
AL="--workload align --steps 3 --warmup 1 --certificate-leg 0 --lanes-leg 0"
DE="--workload decompose --decompose-steps 2 --extra-legs 0"
AP="--workload allpairs --allpairs-steps 2"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python $B $AL > "$OUT/bench_line.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dec_stats" -- python $B $DE > "$OUT/dec_line.json" 2> "$OUT/dec.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ap_stats" -- python $B $AP > "$OUT/ap_line.json" 2> "$OUT/ap.err"
for w in bench dec ap; do
  case $w in bench) ARGS="$AL --steps 2 --cpu-sample 0";; dec) ARGS="$DE --cpu-sample 0";; ap) ARGS="$AP --cpu-sample 0";; esac
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${w}_$c" -- python $B $ARGS > /dev/null 2> "$OUT/pmc_${w}_$c.err"
  done
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_${w}_valu" -- python $B $ARGS > /dev/null 2> "$OUT/pmc_${w}_valu.err"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_default_stats" -- python $B --workload align --steps 3 --warmup 1 > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"
python $B > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
# where the GPU waits for the host: kernel + memory-copy timelines of one step of each pipeline (tools/timeline_gaps.py)
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/tl_dec" -- python $B $DE --cpu-sample 0 > /dev/null 2> "$OUT/tl_dec.err"
python /root/repo/tools/timeline_gaps.py "$OUT/tl_dec" > "$OUT/decompose_timeline_gaps.txt" 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/tl_al" -- python $B $AL --cpu-sample 0 > /dev/null 2> "$OUT/tl_al.err"
python /root/repo/tools/timeline_gaps.py "$OUT/tl_al" encode_codes_kernel > "$OUT/align_timeline_gaps.txt" 2>&1
rm -rf "$OUT/tl_dec" "$OUT/tl_al"
find "$OUT" -name "*.csv" -size +40M -delete
ls "$OUT" | head -60
