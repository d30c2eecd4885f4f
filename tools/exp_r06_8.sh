# round 6, experiment 8: kmer_vote_kernel with its profile rows / window tile requested eight rounds at a time and a branch-free roll
cd /root/repo
cp tracy_amd/lib/libtracy_hip.so /tmp/keep.so
bash tools/ab.sh "python tools/ab_dec.py --extra-legs 0; python tools/ab_align.py 2>&1 | tail -1" vote_base vote_new vote_base vote_new
cp /tmp/keep.so /root/repo/tracy_amd/lib/libtracy_hip.so
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_v -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 --full-line /tmp/full.json > /dev/null 2>&1; python /root/repo/tools/kstats.py /tmp/ks_v 60 | grep -E "kmer_vote|decompose_wave|breakpoint"; grep -o '"pruned[^,]*,' /tmp/full.json | head -3)
timeout 1500 python -m pytest tests/test_gpu_dp.py tests/test_gpu_stream.py tests/test_gpu_parity_slice.py -x -q 2>&1 | tail -5
