import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import tracy_amd
from tracy_amd import hostlib
SC = (3, -5, -10, -4)
ctx = tracy_amd.Context(0)
ctx.set_option("verbose", 1)
for (seed, nt, n, mf) in ((31, 96, 4000, 1000), (5, 256, 3000, 900), (7, 2000, 10000, 1000)):
    refs, profs, rev = hostlib.synth_align(seed, nt, n, mf, 2)
    refl = [r.tobytes() for r in refs]
    for exact in (True, False):
        a = ctx.align_traces(list(profs), refl, SC, 50, 50, exact_scores=exact)
        print(seed, nt, exact, ctx.last_call_stats(), flush=True)
