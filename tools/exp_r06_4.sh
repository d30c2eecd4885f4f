# round 6, experiment 4: alignment rows without divergent branches (rows_base = -DTRACY_ROWS_NO_FAST, rows_fast = the gaps-only form,
# rows_both = the general form select-based as well); the GPU suite runs on the last build
cd /root/repo
cp tracy_amd/lib_ab/rows_both.so /tmp/keep.so
bash tools/ab.sh "python tools/ab_dec.py --extra-legs 0" rows_base rows_fast rows_both rows_base rows_fast rows_both
cp /tmp/keep.so tracy_amd/lib/libtracy_hip.so
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_rows -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 > /dev/null 2>&1; python /root/repo/tools/kstats.py /tmp/ks_rows 40 | grep -E "alignment_rows|homozygous|decompose_wave")
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
