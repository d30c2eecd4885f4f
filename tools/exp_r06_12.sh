# round 6, experiment 12 (host only, on the GPU box's EPYC 9575F): seeding with both ends of the slot pair / bucket requested
# (TRACY_AMD_SEED_ENDS=0: the round's earlier form) and the prefetch distances around the default
cd /root/repo
EXP_THREADS=16 EXP_ENVS="TRACY_AMD_SEED_ENDS=0;TRACY_AMD_SEED_ENDS=1;TRACY_AMD_SEED_ENDS=0;TRACY_AMD_SEED_ENDS=1;TRACY_AMD_SEED_DISTANCE=6,TRACY_AMD_SEED_SLOT_AHEAD=8;TRACY_AMD_SEED_DISTANCE=10,TRACY_AMD_SEED_SLOT_AHEAD=12;TRACY_AMD_SEED_DISTANCE=12,TRACY_AMD_SEED_SLOT_AHEAD=16;TRACY_AMD_SEED_DISTANCE=16,TRACY_AMD_SEED_SLOT_AHEAD=16;TRACY_AMD_SEED_DISTANCE=8,TRACY_AMD_SEED_SLOT_AHEAD=8" timeout 1500 python tools/exp_seed_threads.py 2>&1 | grep traces/s
