#!/bin/bash
# Round profiles (run on the GPU box through gpurun): rocprofv3 kernel stats of the contract benchmark and of the
# decompose benchmark, and the HBM byte counters (separate --pmc passes, kernel trace only) of bench.py.
# The rocprof passes run bench.py's headline leg only (--certificate-leg 0 --lanes-leg 0) so that per-kernel averages and counters are
# not blended with the second leg's smaller launches of the same kernel; one more kernel-trace pass of the DEFAULT
# command is kept too and summarised per (kernel, grid size).
# Outputs under gpurun_out/prof_round/; tools/profile_summarise.py turns them into the files kept in profiles/.
set -u
OUT=/root/repo/gpurun_out/prof_round
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python /root/repo/bench.py --steps 3 --warmup 1 --certificate-leg 0 --lanes-leg 0 > "$OUT/bench_line.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dec_stats" -- python /root/repo/tools/bench_decompose.py --steps 3 --warmup 1 --cpu-sample 32 --extra-legs 0 > "$OUT/dec_line.json" 2> "$OUT/dec.err"
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python /root/repo/bench.py --steps 2 --warmup 1 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0 > /dev/null 2> "$OUT/pmc_$c.err"
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_valu" -- python /root/repo/bench.py --steps 2 --warmup 1 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0 > /dev/null 2> "$OUT/pmc_valu.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_default_stats" -- python /root/repo/bench.py --steps 3 --warmup 1 > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"
python /root/repo/bench.py > "$OUT/bench_plain.json"
python /root/repo/tools/bench_decompose.py --cpu-sample 32 > "$OUT/dec_plain.json" 2>/dev/null
ls -R "$OUT" | head -40
