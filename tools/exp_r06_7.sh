# round 6, experiment 7: where decompose_wave_kernel's time goes -- the kernel cut short after a stage (dw_stopN: -DTRACY_PHASE_CLOCKS
# -DTRACY_DW_NO_ATOMICS -DTRACY_DW_STOP_AFTER=N; results are wrong, only the kernel's duration is read), then the builds compared
cd /root/repo
cp tracy_amd/lib/libtracy_hip.so /tmp/keep.so
cd /tmp && export TMPDIR=/tmp
for n in $STOPS; do
  cp /root/repo/tracy_amd/lib_ab/$n.so /root/repo/tracy_amd/lib/libtracy_hip.so
  rm -rf /tmp/ks_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$n -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 > /dev/null 2>&1
  echo "== $n"; python /root/repo/tools/kstats.py /tmp/ks_$n 60 | grep -E "decompose_wave"
done
cd /root/repo
bash tools/ab.sh "python tools/ab_dec.py --extra-legs 0" $AB
cp /tmp/keep.so /root/repo/tracy_amd/lib/libtracy_hip.so
timeout 1500 python -m pytest tests/test_gpu_decompose.py tests/test_gpu_stream.py tests/test_gpu_parity_slice.py -x -q 2>&1 | tail -5
