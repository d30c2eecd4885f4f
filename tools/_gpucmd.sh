set -x
free -g | head -2; nproc; cat /sys/fs/cgroup/memory.max 2>/dev/null
python bench.py --workload align --steps 10 --warmup 3 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0 > gpurun_out/r05m_align.json 2> gpurun_out/r05m_align.err; tail -c 400 gpurun_out/r05m_align.json
MEMG=$(free -g | awk '/Mem:/{print $7}')
if [ "$MEMG" -gt 80 ]; then
  python bench.py --workload seedextend > gpurun_out/r05m_se.json 2> gpurun_out/r05m_se.err; tail -c 1500 gpurun_out/r05m_se.json; grep VmHWM /proc/self/status
fi
