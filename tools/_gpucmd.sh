set -x
python -m pytest tests/test_gpu_decompose.py tests/test_gpu_stream.py tests/test_gpu_parity_slice.py -x -q -m gpu > gpurun_out/r05d_tests.log 2>&1; tail -15 gpurun_out/r05d_tests.log
python bench.py --workload decompose --decompose-steps 3 --extra-legs 0 --cpu-sample 0 > gpurun_out/r05d_dec.json 2> gpurun_out/r05d_dec.err; tail -c 300 gpurun_out/r05d_dec.json
TRACYHIP_CXXFLAGS=-DTRACY_PHASE_CLOCKS python -c "
from tracy_amd import build as b
b.build(force=True)" > gpurun_out/r05d_build.log 2>&1
python bench.py --workload decompose --decompose-steps 1 --extra-legs 0 --cpu-sample 0 > gpurun_out/r05d_dec_clk.json 2> gpurun_out/r05d_dec_clk.err; grep cycles gpurun_out/r05d_dec_clk.err | tail -3
