# round 6, experiment 9: front_place / front_certify with eight rounds of row entries per wait
cd /root/repo
cp tracy_amd/lib/libtracy_hip.so /tmp/keep.so
bash tools/ab.sh "python tools/ab_dec.py --extra-legs 0; python tools/ab_align.py 2>&1 | tail -1" front_base front_new front_base front_new
cp /tmp/keep.so /root/repo/tracy_amd/lib/libtracy_hip.so
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_f -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 > /dev/null 2>&1; python /root/repo/tools/kstats.py /tmp/ks_f 60 | grep -E "kmer_vote|front_place|front_certify")
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
