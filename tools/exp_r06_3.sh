cd /root/repo
for g in 100000000 8192 4096 16384; do echo "== short grid $g"; TRACYHIP_SHORT_GRID=$g python tools/ab_dec.py --extra-legs 0; done
cd /tmp && export TMPDIR=/tmp
for g in 100000000 8192; do
  TRACYHIP_SHORT_GRID=$g rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$g -- python /root/repo/bench.py --workload decompose --decompose-steps 2 --extra-legs 0 --cpu-sample 0 > /dev/null 2>&1
  echo "== kernel stats, short grid $g"; python /root/repo/tools/kstats.py /tmp/ks_$g 40 | grep -E "alignment_rows|breakpoint|homozygous|front_place|front_certify|rowmax|kmer_vote|af_search|af_prepare|decompose_wave|secdecomp|peaks"
done
TRACYHIP_HOST_TIMERS=1 python /root/repo/bench.py --workload decompose --decompose-steps 9 --extra-legs 0 --cpu-sample 0 2>&1 >/dev/null | grep "^host"
