python bench.py --workload align --steps 20 --warmup 5 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0 > gpurun_out/r05z_al.json 2> gpurun_out/r05z_al.err
python bench.py --workload decompose --decompose-steps 4 --cpu-sample 0 > gpurun_out/r05z_dec.json 2> gpurun_out/r05z_dec.err
python -m pytest tests/test_gpu_stream.py tests/test_gpu_front.py tests/test_gpu_baseline_shapes.py -x -q -m gpu > gpurun_out/r05z_tests.log 2>&1; tail -3 gpurun_out/r05z_tests.log
