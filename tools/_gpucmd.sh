set -x
python -m pytest tests/test_gpu_stream.py tests/test_gpu_decompose.py tests/test_gpu_front.py tests/test_gpu_parity_slice.py -x -q > gpurun_out/r06b_tests.log 2>&1; tail -3 gpurun_out/r06b_tests.log
python bench.py --workload decompose --cpu-sample 0 > gpurun_out/r06b_dec.json 2> gpurun_out/r06b_dec.err; tail -c 600 gpurun_out/r06b_dec.err
TRACYHIP_NO_FRONT_LISTS=1 python bench.py --workload decompose --cpu-sample 0 --extra-legs 0 > gpurun_out/r06b_dec_nolists.json 2> gpurun_out/r06b_dec_nolists.err
python bench.py --workload align --cpu-sample 0 > gpurun_out/r06b_al.json 2> gpurun_out/r06b_al.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python /root/repo/bench.py --workload decompose --cpu-sample 0 --extra-legs 0 --decompose-steps 2 > /dev/null 2> /root/repo/gpurun_out/r06b_prof.err; python /root/repo/tools/timeline_dump.py /tmp/tl > /root/repo/gpurun_out/r06b_timeline.txt
