cd /tmp && export TMPDIR=/tmp
B=/root/repo/bench.py
TRACYHIP_HOST_TIMERS=1 python $B --workload decompose --decompose-steps 9 --extra-legs 0 --cpu-sample 0 2>&1 >/dev/null | grep "^host" 
TRACYHIP_HOST_TIMERS=1 python $B --workload decompose --decompose-traces 12500 --decompose-steps 9 --extra-legs 0 --cpu-sample 0 2>&1 >/dev/null | grep "^host"
TRACYHIP_HOST_TIMERS=1 python $B --workload align --steps 9 --warmup 1 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0 --alone-steps 0 2>&1 >/dev/null | grep "^host"
# effective clock under the sweeps by launch size (GRBM_GUI_ACTIVE / duration): 1, 2.17 and 3 rounds of waves
for n in 4608 10000 13824; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_clock_$n -- python $B --workload align --traces $n --steps 3 --warmup 1 --certificate-leg 0 --lanes-leg 0 --cpu-sample 0 --alone-steps 3 > /dev/null 2>&1
  python - $n <<'PY'
import csv, glob, sys
n = sys.argv[1]
rows = []
for f in glob.glob('/tmp/pmc_clock_%s/**/*counter_collection.csv' % n, recursive=True):
    rows += list(csv.DictReader(open(f)))
big = [r for r in rows if 'gotoh_ckpt_prefix_kernel<15, 16, true' in r['Kernel_Name'] and int(r['Grid_Size']) >= 2 * 64 * int(n)]
if big:
    ns = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in big]
    act = [float(r['Counter_Value']) for r in big]
    print('traces', n, 'launches', len(big), 'avg ms %.3f' % (sum(ns) / len(ns) / 1e6), 'GRBM_GUI_ACTIVE/8/duration GHz', ['%.3f' % (a / 8.0 / d) for a, d in zip(act, ns)][:8])
else:
    print('traces', n, 'no rows', len(rows), rows[0].keys() if rows else '')
PY
done
