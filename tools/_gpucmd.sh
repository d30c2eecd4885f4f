python -m pytest tests/test_gpu_band16.py tests/test_gpu_dp.py tests/test_gpu_stream.py tests/test_gpu_decompose.py tests/test_gpu_baseline_shapes.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r05t_tests.log 2>&1; tail -3 gpurun_out/r05t_tests.log
python bench.py --workload decompose --decompose-steps 3 --extra-legs 0 --cpu-sample 0 > gpurun_out/r05t_dec.json 2> gpurun_out/r05t_dec.err
python bench.py --workload align --steps 10 --warmup 3 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0 > gpurun_out/r05t_al.json 2> gpurun_out/r05t_al.err
cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d /root/repo/gpurun_out/r05t_$c -- python /root/repo/bench.py --workload decompose --decompose-steps 1 --extra-legs 0 --cpu-sample 0 > /dev/null 2> /root/repo/gpurun_out/r05t_pmc.err
python /root/repo/tools/pmc_by_kernel.py /root/repo/gpurun_out/r05t_$c band16 > /root/repo/gpurun_out/r05t_$c.txt
rm -rf /root/repo/gpurun_out/r05t_$c
done
