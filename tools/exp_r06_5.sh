# round 6, experiment 5: findBreakpoint with the profile rows of eight rounds of columns requested together and one window sum per
# position (bp_base / bp_new on one box), its tests, and decompose_wave_kernel's cycles by stage (-DTRACY_PHASE_CLOCKS)
cd /root/repo
cp tracy_amd/lib/libtracy_hip.so /tmp/keep.so
bash tools/ab.sh "python tools/ab_dec.py --extra-legs 0" bp_base bp_new bp_base bp_new
cp tracy_amd/lib_ab/dw_clocks.so tracy_amd/lib/libtracy_hip.so
python bench.py --workload decompose --decompose-steps 1 --warmup 1 --extra-legs 0 --cpu-sample 0 2>&1 | grep "cycles per trace" | tail -3
cp /tmp/keep.so tracy_amd/lib/libtracy_hip.so
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_bp -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 > /dev/null 2>&1; python /root/repo/tools/kstats.py /tmp/ks_bp 40 | grep -E "alignment_rows|breakpoint|decompose_wave")
timeout 1500 python -m pytest tests/test_gpu_decompose.py tests/test_gpu_stream.py tests/test_gpu_parity_slice.py tests/test_gpu_cli.py -x -q 2>&1 | tail -5
