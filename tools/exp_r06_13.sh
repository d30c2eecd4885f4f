# round 6, experiment 13: the decompose call's own planning loops inside plan_common's two passes (PlanHooks) instead of two passes of their own
cd /root/repo
cp tracy_amd/lib/libtracy_hip.so /tmp/keep.so
bash tools/ab.sh "TRACYHIP_HOST_TIMERS=1 python tools/ab_dec.py --extra-legs 0 2>&1 | grep -E '^dec|stream_decompose.plan '" plan_base plan_new plan_base plan_new
cp /tmp/keep.so /root/repo/tracy_amd/lib/libtracy_hip.so
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
