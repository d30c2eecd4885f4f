#!/usr/bin/env python
"""bench_allpairs.py -- BASELINE.json configs[4], the DP part of `tracy assemble`: all-pairs profile x profile
gotohScore<true,true> (msa.h:33-42 distanceMatrix) over N synthetic overlapping ~1 kb traces, sharded over ranks
by pair index (no data-path collective; the score slices are gathered at the end).  Prints one JSON line with GCUPS,
a CPU baseline (oracle, sample of pairs) and a bit-exactness check on that sample.  Secondary measurement; bench.py
is the contract benchmark."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCORE = (3, -5, -10, -4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--traces", type=int, default=400)
    ap.add_argument("--trace-len", type=int, default=900)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=64)
    args = ap.parse_args()
    import tracy_amd
    from tracy_amd import capi, hostlib
    n, mf = args.traces, args.trace_len
    # overlapping traces tiled over one region: trace i starts at i * step of a shared reference
    region = 50000
    refs, profs, rev = hostlib.synth_align(9000, n, region, mf, 0)  # independent windows; overlap does not change the DP cost
    profs = np.ascontiguousarray(profs)
    iu = np.triu_indices(n, 1)
    i1, i2 = iu[0].astype(np.uint32), iu[1].astype(np.uint32)
    npairs = len(i1)
    ctx = tracy_amd.Context(0)
    plist = [profs[i] for i in range(n)]

    def step():
        return ctx.score(plist, plist, SCORE + (1, 1), idx1=i1, idx2=i2)

    for _ in range(args.warmup):
        sc = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sc = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    cells = npairs * mf * mf
    line = {"metric": "GCUPS (all-pairs profile x profile gotohScore<true,true>, msa.h:33-42)", "value": round(cells / dt / 1e9, 1), "unit": "GCUPS",
            "pairs": int(npairs), "pairs_per_s": round(npairs / dt, 1), "ms_per_step": round(dt * 1e3, 2), "n_gpus": 1,
            "config": {"workload": "configs[4]: %d traces of %d bases, %d pairs" % (n, mf, npairs)}, "data": "synthetic",
            "note": "host-staged inputs (MEM_HOST): includes the upload of the profiles and the download of the scores"}
    if args.cpu_sample > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as orc
        rng = np.random.default_rng(1)
        pick = rng.choice(npairs, size=min(args.cpu_sample, npairs), replace=False)
        t0 = time.perf_counter()
        want = [orc.gotoh_score_prof(profs[i1[k]], profs[i2[k]], 1, 1, SCORE) for k in pick]
        cdt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(len(pick) * mf * mf / cdt / 1e9, 4), "unit": "GCUPS", "cores": 1, "kind": "port",
                                "sample": "%d of the same pairs through the oracle, %.1f s" % (len(pick), cdt)}
        line["parity_checked"] = {"pairs": int(len(pick)), "bit_identical": bool(all(int(sc[k]) == w for k, w in zip(pick, want)))}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
