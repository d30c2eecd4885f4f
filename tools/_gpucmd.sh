set -x
python -m pytest tests -x -q -m gpu > gpurun_out/r05k_tests.log 2>&1; tail -15 gpurun_out/r05k_tests.log
python bench.py > gpurun_out/r05k_bench.json 2> gpurun_out/r05k_bench.err; tail -c 300 gpurun_out/r05k_bench.json
