python -m pytest tests/test_gpu_dp.py tests/test_gpu_stream.py tests/test_gpu_decompose.py tests/test_gpu_front.py tests/test_gpu_baseline_shapes.py tests/test_gpu_cli.py tests/test_gpu_parity_slice.py -x -q -m gpu > gpurun_out/r06a_tests.log 2>&1; tail -3 gpurun_out/r06a_tests.log
python bench.py --workload decompose --decompose-steps 4 --cpu-sample 0 > gpurun_out/r06a_dec.json 2> gpurun_out/r06a_dec.err
python bench.py --workload align --steps 10 --warmup 3 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0 > gpurun_out/r06a_al.json 2> gpurun_out/r06a_al.err
