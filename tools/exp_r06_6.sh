# round 6, experiment 6: decompose_wave_kernel with its loads requested together / branch-free, four words per pass in the sets and scans
cd /root/repo
cp tracy_amd/lib/libtracy_hip.so /tmp/keep.so
bash tools/ab.sh "python tools/ab_dec.py --extra-legs 0" dw_base dw_new dw_base dw_new
cp tracy_amd/lib_ab/dw_clocks.so tracy_amd/lib/libtracy_hip.so
python bench.py --workload decompose --decompose-steps 1 --warmup 1 --extra-legs 0 --cpu-sample 0 2>&1 | grep "cycles per trace" | tail -2
cp /tmp/keep.so tracy_amd/lib/libtracy_hip.so
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_dw -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 > /dev/null 2>&1; python /root/repo/tools/kstats.py /tmp/ks_dw 40 | grep -E "alignment_rows|breakpoint|decompose_wave")
timeout 1500 python -m pytest tests/test_gpu_decompose.py tests/test_gpu_stream.py tests/test_gpu_parity_slice.py -x -q 2>&1 | tail -5
