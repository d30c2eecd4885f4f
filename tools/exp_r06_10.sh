# round 6, experiment 10 (no gain, reverted: trace / origin / pruned-sweep timers 13.3 / 5.03 / 11.2 vs 13.5 / 5.05 / 11.3 ms): band kernels stage
# their reference codes by dwords (four rounds of the lanes per wait) and their kept rows
# branch-free
cd /root/repo
cp tracy_amd/lib/libtracy_hip.so /tmp/keep.so
bash tools/ab.sh "python tools/ab_dec.py; python tools/ab_align.py 2>&1 | tail -1" stage_base stage_new stage_base stage_new
cp /tmp/keep.so /root/repo/tracy_amd/lib/libtracy_hip.so
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_s -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 > /dev/null 2>&1; python /root/repo/tools/kstats.py /tmp/ks_s 60 | grep -E "band16")
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
