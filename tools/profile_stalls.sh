#!/bin/bash
# Stall counters of the tracy decompose step (and the align step), per kernel: two --pmc passes with --kernel-trace only.
#   pass A: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU
#   pass B: SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES
# Output under gpurun_out/stalls_<tag>/; tools/pmc_by_kernel.py prints the per-kernel sums.
set -u
TAG=${1:-r05}
OUT=/root/repo/gpurun_out/stalls_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B=/root/repo/bench.py
DE="--workload decompose --decompose-steps 1 --extra-legs 0 --cpu-sample 0"
AL="--workload align --steps 2 --warmup 1 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0"
for w in dec al; do
  case $w in dec) ARGS="$DE";; al) ARGS="$AL";; esac
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU \
    --output-format csv -d "$OUT/${w}_A" -- python $B $ARGS > /dev/null 2> "$OUT/${w}_A.err"
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d "$OUT/${w}_B" -- python $B $ARGS > /dev/null 2> "$OUT/${w}_B.err"
  for p in A B; do
    python /root/repo/tools/pmc_by_kernel.py "$OUT/${w}_$p" > "$OUT/${w}_$p.txt" 2>&1
  done
done
find "$OUT" -name "*.csv" -size +20M -delete
ls -la "$OUT"
