python -m pytest tests -x -q -m gpu > gpurun_out/r05x_tests.log 2>&1; tail -3 gpurun_out/r05x_tests.log
python bench.py --workload decompose --decompose-steps 4 --cpu-sample 0 > gpurun_out/r05x_dec.json 2> gpurun_out/r05x_dec.err
