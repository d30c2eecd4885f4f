#!/bin/bash
# scratch experiment script (GPU box): tests, decompose A/B of the pruned sweep, stall counters of the band kernels
set -u
OUT=/root/repo/gpurun_out/x1
rm -rf "$OUT"; mkdir -p "$OUT"
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; echo "tests rc=$?" >> "$OUT/tests.log"
TRACYHIP_HOST_TIMERS=1 python bench.py --workload decompose --decompose-steps 3 --cpu-sample 0 > "$OUT/dec_front.json" 2> "$OUT/dec_front.err"
TRACYHIP_NO_FRONT=1 python bench.py --workload decompose --decompose-steps 3 --cpu-sample 0 --extra-legs 0 > "$OUT/dec_nofront.json" 2> "$OUT/dec_nofront.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1
DE="--workload decompose --decompose-steps 1 --extra-legs 0 --cpu-sample 0 --decompose-traces 40000"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$OUT/pmcA" -- python /root/repo/bench.py $DE > /dev/null 2> "$OUT/pmcA.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d "$OUT/pmcB" -- python /root/repo/bench.py $DE > /dev/null 2> "$OUT/pmcB.err"
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_IFETCH --output-format csv -d "$OUT/pmcC" -- python /root/repo/bench.py $DE > /dev/null 2> "$OUT/pmcC.err"
# keep only the counter csvs (small)
find "$OUT" -name "*.csv" -size +20M -delete
ls -R "$OUT" | head -50
