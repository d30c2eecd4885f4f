#!/usr/bin/env python
"""experiment: what the windowed multi-pass form of the final tracebacks would cost.
Times the traceback kernel on (a) 10 000 profile x string pairs of 1000 x 1020 (today's final alignments, K = 16, one pass)
and (b) the same number of cells cut into the pieces the certified band leaves: 40 000 pairs of 256 rows x 352 columns
(K = 4, passes of 256 rows, window 256 + 2 W, W = 48).  Kernel time from the library's timers; no boundary hand-over
between the pieces, so (b) is a lower bound of the real thing by the hand-over rows only."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tracy_amd
from tracy_amd import capi

def run(ctx, lib, npairs, m, n, label):
    rng = np.random.default_rng(1)
    base = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n)
    refs = [base.tobytes()] * 1
    p = np.zeros((6, m), dtype=np.float32)
    idx = {65: 0, 67: 1, 71: 2, 84: 3}
    for j in range(m):
        p[:4, j] = 0.04
        p[idx[int(base[min(j, n - 1)])], j] = 0.88
    a1 = [p]
    i1 = [0] * npairs
    i2 = [0] * npairs
    ctx.align(a1, refs, (3, -5, -10, -4, 1, 0), idx1=i1[:64], idx2=i2[:64])
    lib.tracyhip_timing_enable(ctx._h, 1)
    lib.tracyhip_timing_reset(ctx._h)
    ctx.align(a1, refs, (3, -5, -10, -4, 1, 0), idx1=i1, idx2=i2)
    kt = capi.KernelTiming()
    lib.tracyhip_timing_get(ctx._h, 1, C.byref(kt))
    lib.tracyhip_timing_enable(ctx._h, 0)
    print("%s: %d pairs of %d x %d: traceback kernel %.2f ms in %d launch(es), %.1f GCUPS" % (label, npairs, m, n, kt.ms, kt.launches, npairs * m * n / kt.ms / 1e6))

ctx = tracy_amd.Context(0)
lib = capi.lib()
run(ctx, lib, 10000, 1000, 1020, "today ")
run(ctx, lib, 40000, 256, 352, "pieces")
run(ctx, lib, 80000, 128, 224, "pieces (K = 4 on half the lanes)")
ctx.close()
