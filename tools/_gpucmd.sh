python -m pytest tests/test_gpu_decompose.py tests/test_gpu_stream.py -x -q -m gpu > gpurun_out/r05s_tests.log 2>&1; tail -3 gpurun_out/r05s_tests.log
python bench.py --workload decompose --decompose-steps 3 --cpu-sample 0 > gpurun_out/r05s_dec.json 2> gpurun_out/r05s_dec.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r05s_stats -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 > /dev/null 2> /root/repo/gpurun_out/r05s_prof.err
find /root/repo/gpurun_out/r05s_stats -name "*kernel_trace.csv" -delete
