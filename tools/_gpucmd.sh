set -x
python -m pytest tests/test_gpu_decompose.py tests/test_gpu_stream.py tests/test_gpu_parity_slice.py -x -q -m gpu > gpurun_out/r05j_tests.log 2>&1; tail -5 gpurun_out/r05j_tests.log
python bench.py --workload decompose --decompose-steps 3 --extra-legs 0 --cpu-sample 0 > gpurun_out/r05j_dec.json 2> gpurun_out/r05j_dec.err; tail -c 200 gpurun_out/r05j_dec.json
TRACYHIP_NO_STREAM_PRIORITY=1 python bench.py --workload decompose --decompose-steps 3 --extra-legs 0 --cpu-sample 0 > gpurun_out/r05j_dec_noprio.json 2> gpurun_out/r05j_dec_noprio.err; tail -c 200 gpurun_out/r05j_dec_noprio.json
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/r05j_tl -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 > /dev/null 2> /root/repo/gpurun_out/r05j_prof.err
python /root/repo/tools/timeline_dump.py /root/repo/gpurun_out/r05j_tl > /root/repo/gpurun_out/r05j_timeline.txt 2>&1
rm -rf /root/repo/gpurun_out/r05j_tl
