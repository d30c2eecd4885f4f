cd /tmp && export TMPDIR=/tmp
B=/root/repo/bench.py
AL="--workload align --steps 4 --warmup 1 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0 --alone-steps 0"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_al -- python $B $AL > /dev/null 2>&1
python /root/repo/tools/timeline_dump.py /tmp/tl_al > /root/repo/gpurun_out/r06d_align_timeline.txt 2>&1
for n in 4608 9216 10000 13824; do
  python $B --workload align --traces $n --steps 5 --warmup 2 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('traces', d['config']['traces_per_gpu'], 'ms', d['ms_per_step'], 'score_ms', r['ms_score'], 'frac', r['frac'], 'alone_ms', r['dominant_kernel_alone_ms'], 'alone_frac', r['dominant_kernel_alone_frac'])"
done
TRACYHIP_HOST_TIMERS=1 python $B --workload decompose --decompose-steps 3 --extra-legs 0 --cpu-sample 0 2>&1 >/dev/null | grep -v "band stage\|stream-ordered" | tail -30
