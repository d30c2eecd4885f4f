import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import tracy_amd
from tracy_amd import hostlib, capi
SC = (3, -5, -10, -4)
ctx = tracy_amd.Context(0)
ctx.set_option("verbose", 1)
for (seed, nd, n, mf) in ((4711, 160, 3000, 1000), (12, 2000, 3000, 1000)):
    d = hostlib.synth_decompose_batch(seed, nd, n, mf, 0, mix=1)
    refs = [d["refs"][i].tobytes() for i in range(nd)]
    for exact in (True, False):
        hbc = capi.HostBaseCalls([d["signal"][i] for i in range(nd)], [d["bcpos"][i] for i in range(nd)],
                                 [d["primary"][i].tobytes() for i in range(nd)], [d["secondary"][i].tobytes() for i in range(nd)])
        a = ctx.decompose_traces([d["profiles"][i] for i in range(nd)], hbc, refs, SC, exact_scores=exact)
        print(seed, nd, exact, ctx.last_call_stats(), flush=True)
ctx.close()
