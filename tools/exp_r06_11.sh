# round 6, experiment 11 (no gain, not kept: 0.915 vs 0.89-0.90 ms): alignment rows request the next batch's op bytes before waiting for this batch's reference bytes
cd /root/repo
cp tracy_amd/lib/libtracy_hip.so /tmp/keep.so
bash tools/ab.sh "python tools/ab_dec.py --extra-legs 0" rowsp_base rowsp_new rowsp_base rowsp_new
cp /tmp/keep.so /root/repo/tracy_amd/lib/libtracy_hip.so
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_r -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 > /dev/null 2>&1; python /root/repo/tools/kstats.py /tmp/ks_r 60 | grep -E "alignment_rows")
