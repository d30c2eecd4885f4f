"""How the 16-bit score sweep (tracyhip_gotoh_score, profile x reference, 900 x 10 000) fills the device: time against the number
of pairs -- rounds of resident waves, the last wave of a SIMD on its own (development tool, not bench.py).
   python tools/exp_sweep_rounds.py [m] [n] [npairs ...]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import tracy_amd
from tracy_amd import capi


def main(m, n, counts):
    rng = np.random.default_rng(0)
    nmax = max(counts)
    base_ref = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n + nmax).astype(np.uint8)
    refs = [base_ref[i:i + n].tobytes() for i in range(nmax)]
    x = rng.random((4, m + nmax)).astype(np.float32) ** 6
    x /= x.sum(axis=0, keepdims=True)
    profs = []
    for i in range(nmax):
        p = np.zeros((6, m), dtype=np.float32)
        p[:4] = x[:, i:i + m]
        profs.append(p)
    ctx = tracy_amd.Context(0)
    lib = capi.lib()
    prm = capi.Params(3, -5, -10, -4, 1, 0)
    for npairs in counts:
        a1 = capi.PackedSeqs(profs[:npairs])
        a2 = capi.PackedSeqs(refs[:npairs])
        d1 = torch.from_numpy(a1.data).cuda()
        d2 = torch.from_numpy(a2.data).cuda()
        scores = torch.zeros(npairs, dtype=torch.int32, device="cuda")
        pr = capi.Pairs()
        pr.npairs = npairs
        pr.a1 = a1.seqset(d1.data_ptr())
        pr.a2 = a2.seqset(d2.data_ptr())
        best = 1e9
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = lib.tracyhip_gotoh_score(ctx._h, C.byref(pr), C.byref(prm), capi.MEM_DEVICE, C.c_void_p(scores.data_ptr()))
            assert rc == 0, lib.tracyhip_last_error()
            torch.cuda.synchronize()
            if it:
                best = min(best, time.perf_counter() - t0)
        print("pairs %6d  %8.3f ms  %7.1f GCUPS  %6.3f ms per 1024 pairs" % (npairs, best * 1e3, npairs * m * n / best / 1e9, best * 1e3 / npairs * 1024), flush=True)


if __name__ == "__main__":
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 900
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    counts = [int(v) for v in sys.argv[3:]] or [256, 1024, 2048, 3072, 4096, 5120, 6144, 8192, 10000, 10240, 12288, 15360, 20480]
    main(m, n, counts)
