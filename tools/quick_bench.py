"""Rough device-resident timing of the primitive DP entry points (development tool, not bench.py)."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import tracy_amd
from tracy_amd import capi


def main(npairs=2048, m=900, n=10000, mode="qp"):
    rng = np.random.default_rng(0)
    refs = [bytes(rng.choice(list(b"ACGT"), size=n).tolist()) for _ in range(npairs)]
    if mode == "qp":
        x = rng.random((npairs, 4, m)).astype(np.float32) ** 6
        profs = np.zeros((npairs, 6, m), dtype=np.float32)
        profs[:, :4] = x / x.sum(axis=1, keepdims=True)
        a1 = capi.PackedSeqs(list(profs))
    else:
        a1 = capi.PackedSeqs([bytes(rng.choice(list(b"ACGT"), size=m).tolist()) for _ in range(npairs)])
    a2 = capi.PackedSeqs(refs)
    ctx = tracy_amd.Context(0)
    d1 = torch.from_numpy(a1.data).cuda()
    d2 = torch.from_numpy(a2.data).cuda()
    scores = torch.zeros(npairs, dtype=torch.int32, device="cuda")
    off = (np.arange(npairs, dtype=np.uint64) * np.uint64(m + n))
    ops = torch.zeros(npairs * (m + n), dtype=torch.uint8, device="cuda")
    olen = torch.zeros(npairs, dtype=torch.int32, device="cuda")
    pr = capi.Pairs()
    pr.npairs = npairs
    pr.a1 = a1.seqset(d1.data_ptr())
    pr.a2 = a2.seqset(d2.data_ptr())
    prm = capi.Params(3, -5, -10, -4, 1, 0)
    lib = capi.lib()
    cells = npairs * m * n
    for name in ("score", "align"):
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if name == "score":
                rc = lib.tracyhip_gotoh_score(ctx._h, C.byref(pr), C.byref(prm), capi.MEM_DEVICE, C.c_void_p(scores.data_ptr()))
            else:
                rc = lib.tracyhip_gotoh_align(ctx._h, C.byref(pr), C.byref(prm), capi.MEM_DEVICE, C.c_void_p(scores.data_ptr()),
                                              C.c_void_p(ops.data_ptr()), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                              C.c_void_p(olen.data_ptr()))
            assert rc == 0, lib.tracyhip_last_error()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("%s %s: %.1f ms  %.1f GCUPS  (scores[0..3]=%s)" % (mode, name, dt * 1e3, cells / dt / 1e9, scores[:3].tolist()), flush=True)


if __name__ == "__main__":
    np_ = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    main(np_, mode="qp")
    main(np_, mode="char")
