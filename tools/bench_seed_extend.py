#!/usr/bin/env python
"""bench_seed_extend.py -- BASELINE.json configs[3] in miniature: traces sampled from a synthetic genome, host k-mer
seeding (getReferenceSlice, all host threads) + the device extend (tracyhip_align_traces with job.oriented: one
score pass with checkpoints, band traceback, trimReferenceSlice, final alignment).  Prints one JSON line with the
seeding rate (traces/s, host), the extend rate (traces/s and GCUPS, device) and the end-to-end rate.  chr22 is not
available offline: the genome is synthetic (stated in `data`).  Secondary measurement."""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCORE = (3, -5, -10, -4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--traces", type=int, default=4000)
    ap.add_argument("--genome-mb", type=float, default=20.0)
    ap.add_argument("--trace-len", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    import tracy_amd
    from tracy_amd import hostlib
    rng = np.random.default_rng(22)
    n = int(args.genome_mb * 1e6)
    genome = rng.integers(0, 4, size=n, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = lut[genome].tobytes()
    tmp = tempfile.mkdtemp()
    gpath = os.path.join(tmp, "genome.fa.gz")
    with gzip.open(gpath, "wb", compresslevel=1) as f:
        f.write(b">chrSyn\n" + seq + b"\n")
    t0 = time.perf_counter()
    g = hostlib.Genome(gpath, 15, 0)
    t_index = time.perf_counter() - t0
    # traces: basecalled synthetic chromatograms of genome stretches (1 % substitutions), half reverse strand
    nt, mf = args.traces, args.trace_len
    refs = np.zeros((nt, mf + 40), np.uint8)
    starts = rng.integers(0, n - mf - 50, size=nt)
    cons, profs = [], []
    # reuse the workload generator of bench.py for the chromatogram + basecall + profile: windows of exactly mf+40 bases
    win = np.stack([np.frombuffer(seq[s:s + mf + 40], dtype=np.uint8) for s in starts])
    profs_arr = np.zeros((nt, 6, mf), np.float32)
    for i in range(nt):
        s = win[i].tobytes()[:mf]
        if i % 2:
            s = s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
        p = np.zeros((6, mf), np.float32)
        idx = np.frombuffer(s, np.uint8)
        code = np.searchsorted(lut, idx)
        err = rng.random(mf) < 0.01
        code = np.where(err, (code + 1) % 4, code)
        p[:4] = 0.02
        p[code, np.arange(mf)] = 0.94
        profs_arr[i] = p
        cons.append(lut[code].tobytes())
    t0 = time.perf_counter()
    sd = g.seed(cons, 50, 50, 3, 1000, 0)
    t_seed = time.perf_counter() - t0
    ok = np.nonzero(sd["status"] == 1)[0]
    ctx = tracy_amd.Context(0)
    plist = [profs_arr[i] for i in ok]
    wins = [sd["slices"][i] for i in ok]
    fwd = [int(sd["forward"][i]) for i in ok]
    from tracy_amd import capi
    pp, pw = capi.PackedSeqs(plist, capi.SEQ_PROFILE), capi.PackedSeqs(wins, capi.SEQ_CHAR)  # packed host buffers, as a C caller holds them
    ctx.align_traces(pp, pw, SCORE, 50, 50, oriented=fwd)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = ctx.align_traces(pp, pw, SCORE, 50, 50, oriented=fwd)
    t_ext = (time.perf_counter() - t0) / args.steps
    cells = sum((mf - 100) * len(w) for w in wins) * 2 + int((mf * res["slice_len"].astype(np.int64)).sum())
    right = int(sum(1 for k, i in enumerate(ok) if abs(int(sd["pos"][i]) + int(res["ref_pos"][k]) - (int(starts[i]) - (50 if not i % 2 else 0))) <= 60))
    line = {"metric": "traces/s (host k-mer seeding + device Gotoh extend, configs[3] in miniature)",
            "value": round(len(ok) / (t_seed + t_ext), 1), "unit": "traces/s", "n_gpus": 1,
            "seed_traces_per_s": round(nt / t_seed, 1), "seed_threads": os.cpu_count(), "usable_cores": int(hostlib.lib().tracyhost_usable_threads()), "extend_traces_per_s": round(len(ok) / t_ext, 1),
            "extend_gcups": round(cells / t_ext / 1e9, 1), "index_build_s": round(t_index, 2), "anchored": int(len(ok)), "traces": nt,
            "placed_within_60bp_of_truth": right,
            "config": {"workload": "%d traces of %d bases vs a %.0f Mb synthetic genome, k=15, window = trace + 2*1000" % (nt, mf, args.genome_mb)},
            "data": "synthetic (GRCh38 chr22 is not available offline)",
            "note": "extend uses host-staged buffers (MEM_HOST): upload of profiles/windows and download of results included"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
