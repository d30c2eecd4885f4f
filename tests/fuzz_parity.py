#!/usr/bin/env python
"""fuzz_parity.py -- randomized GPU-vs-oracle campaign (development tool, run through gpurun): ragged random pairs in
every DP mode / AlignConfig / scoring, ragged `tracy align` batches, `tracy decompose` batches.  Prints one JSON line
with the number of cases compared and the mismatches (0 expected); exits non-zero on any mismatch."""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def rand_seq(rng, n, alpha=b"ACGT"):
    return bytes(rng.choice(list(alpha), size=n).tolist())


def rand_profile(rng, n, kind):
    p = np.zeros((6, n), np.float32)
    x = rng.random((4, n)).astype(np.float32)
    if kind == 0:
        x = x ** 8
    p[:4] = x / x.sum(axis=0, keepdims=True)
    if kind == 2 and n:  # alignment-like profile with N / gap weight
        j = rng.integers(0, n, size=max(1, n // 10))
        p[4, j] = np.float32(0.25)
        p[5, j] = np.float32(0.125)
    if kind == 3 and n:  # columns whose scores sit on integers: one-hot, uniform, sixteenths (the screened profile score's hard cases)
        j = rng.integers(0, n, size=max(1, n // 3))
        p[:4, j] = 0
        p[rng.integers(0, 4, size=len(j)), j] = 1
        j = rng.integers(0, n, size=max(1, n // 20))
        p[:4, j] = np.float32(0.25)
        j = rng.integers(0, n, size=max(1, n // 5))
        p[:4, j] = (rng.multinomial(16, [0.25] * 4, size=len(j)).T / 16.0).astype(np.float32)
    return p


def related(rng, s, rate=0.1):
    out = bytearray()
    for ch in s:
        u = rng.random()
        if u < rate / 3:
            continue
        if u < 2 * rate / 3:
            out.append(int(rng.choice(list(b"ACGT"))))
        out.append(int(rng.choice(list(b"ACGT"))) if u > 1 - rate / 3 else ch)
    return bytes(out)


def run_campaign(pairs=1500, traces=96, lanes=1, seed=1, decompose_len=(700, 2200), decompose_mix=1):
    """one campaign; returns {"compared": {...}, "mismatches": n, "first": [...]} (also what main() prints)"""
    args = argparse.Namespace(pairs=pairs, traces=traces, lanes=lanes, seed=seed)
    import pyoracle as orc
    import tracy_amd
    from tracy_amd import capi, hostlib
    import sage_oracle as so
    import indigo_oracle as io
    rng = np.random.default_rng(args.seed)
    ctx = tracy_amd.Context(0)
    ctx.set_lanes(max(1, args.lanes))
    bad, done = [], {}
    pool = ThreadPoolExecutor(max_workers=16)

    # ---- 1. DP modes ----
    scorings = [(3, -5, -10, -4), (5, -4, -10, -1), (1, -1, -2, -1), (2, -3, 0, -2), (7, -9, -30, -3)]
    for mode in ("char", "qp", "prof"):
        for cfg in [(1, 0), (1, 1), (0, 0), (0, 1)]:
            sc = scorings[int(rng.integers(0, len(scorings)))]
            n_pairs = args.pairs // 12
            a1, a2 = [], []
            for _ in range(n_pairs):
                m = int(rng.choice([0, 1, 2, 63, 64, 65, rng.integers(1, 700), rng.integers(1, 1500)]))
                n = int(rng.choice([0, 1, 3, 64, rng.integers(1, 900)]))
                if mode == "char":
                    s2 = rand_seq(rng, n, b"ACGTN")
                    s1 = (related(rng, s2) + rand_seq(rng, m))[:m] if rng.random() < 0.7 else rand_seq(rng, m)
                    a1.append(s1); a2.append(s2)
                elif mode == "qp":
                    a1.append(rand_profile(rng, m, int(rng.choice([0, 1, 3])))); a2.append(rand_seq(rng, n, b"ACGTACGTNn-x" if rng.random() < 0.5 else b"ACGT"))
                else:
                    m, n = min(m, 400), min(n, 400)
                    a1.append(rand_profile(rng, m, int(rng.integers(0, 4)))); a2.append(rand_profile(rng, n, int(rng.integers(0, 4))))
            scores, btr = ctx.align(a1, a2, sc + cfg)
            sonly = ctx.score(a1, a2, sc + cfg)

            def want(i):
                if mode == "char":
                    return orc.gotoh_str(a1[i], a2[i], cfg[0], cfg[1], sc)
                p2 = orc.create_profile_str(a2[i]) if mode == "qp" else a2[i]
                return orc.gotoh_prof(a1[i], p2, cfg[0], cfg[1], sc)
            for i, w in enumerate(pool.map(want, range(n_pairs))):
                if (int(scores[i]), btr[i]) != w or int(sonly[i]) != w[0]:
                    bad.append(("dp", mode, cfg, sc, i, len(a1[i]) if mode == "char" else a1[i].shape[1]))
            done["dp_" + mode] = done.get("dp_" + mode, 0) + n_pairs
            if mode == "char" and cfg[1] == 0:
                # the band kernels through tracyhip_gotoh_banded: a band as wide as the oracle's score allows gap steps holds
                # every optimal path, so the banded result is the whole-matrix result
                sub, lo, hi = [], [], []
                for i in range(n_pairs):
                    m, n = len(a1[i]), len(a2[i])
                    if m < 1 or n < 1:
                        continue
                    g = (sc[0] * min(m, n) - int(scores[i])) // -sc[3] + 1
                    if cfg[0]:
                        fwd = btr[i][::-1]
                        d1 = n - (len(fwd) - len(fwd.rstrip(b"h"))) - m
                        dl, dh = d1 - g - 1, d1 + g + 1
                    else:
                        dl, dh = min(0, n - m) - g - 1, max(0, n - m) + g + 1
                    if dh - dl <= 175:
                        sub.append(i); lo.append(dl); hi.append(dh)
                if sub:
                    bs, bb = ctx.align_banded([a1[i] for i in sub], [a2[i] for i in sub], sc + cfg, lo, hi)
                    for j, i in enumerate(sub):
                        if (int(bs[j]), bb[j]) != (int(scores[i]), btr[i]):
                            bad.append(("banded", cfg, sc, i, len(a1[i]), len(a2[i]), lo[j], hi[j]))
                    done["dp_banded"] = done.get("dp_banded", 0) + len(sub)

    # ---- 2. `tracy align` batches, ragged ----
    nt = args.traces
    profs, refs = [], []
    long_traces = args.seed % 5 == 0  # every fifth campaign: traces beyond one pass of the tallest strips (1024 rows) and the 16-bit range checks
    for i in range(nt):
        mf = int(rng.integers(1000, 2500)) if long_traces else int(rng.integers(120, 1100))
        n = int(rng.integers(mf + 50, mf + 3000 if long_traces else 4000))
        r, p, _ = hostlib.synth_align(int(rng.integers(0, 1 << 30)), 1, n, mf, 1)
        if rng.random() < 0.2:  # windows with N columns take the six-code form of the sweeps, the others the compact one
            r = r.copy()
            r[0, rng.integers(0, n, size=int(rng.integers(1, 6)))] = ord("N")
        profs.append(p[0]); refs.append(r[0].tobytes())
    # the scoring of the pipelines: tracy's default in two campaigns of three, otherwise one whose gap costs change what the certificates
    # of the pruned sweeps and the bands can use (ge = -1: no second certificate; cheap gaps: wide bands; dear gaps: narrow ones)
    psc = (3, -5, -10, -4) if args.seed % 3 else [(5, -4, -10, -1), (2, -3, -5, -2), (1, -1, -2, -1), (4, -6, -20, -8)][(args.seed // 3) % 4]
    for (tl, tr) in [(50, 50), (0, 0), (13, 77)]:
        got = ctx.align_traces(profs, refs, psc, tl, tr)
        for i, w in enumerate(pool.map(lambda i: so.align_trace(profs[i], refs[i], psc, tl, tr), range(nt))):
            ok = all(int(got[k][i]) == int(w[k]) for k in ("score_fwd", "score_rev", "forward", "score_prelim", "slice_begin", "slice_len", "ref_pos",
                                                            "score_final")) and got["btr"][i] == w["btr"]
            if not ok:
                bad.append(("align", tl, tr, i))
        done["align"] = done.get("align", 0) + nt
        # the library's default mode (strand by certificate): same decision and alignment; the winner's orientation
        # score is exact, the loser's is its exact score or a certified upper bound of it
        fast = ctx.align_traces(profs, refs, psc, tl, tr, exact_scores=False)
        for i in range(nt):
            ok = all(int(fast[k][i]) == int(got[k][i]) for k in ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"))
            ok = ok and fast["btr"][i] == got["btr"][i]
            win, lose = ("score_fwd", "score_rev") if int(got["forward"][i]) else ("score_rev", "score_fwd")
            ok = ok and int(fast[win][i]) == int(got[win][i]) and int(fast[lose][i]) >= int(got[lose][i])
            if not ok:
                bad.append(("align_certificate", tl, tr, i))
        done["align_certificate"] = done.get("align_certificate", 0) + nt

    # ---- 3. `tracy decompose` batches ----
    nd = max(8, args.traces // 2)
    mf, n = decompose_len
    if long_traces and decompose_len == (700, 2200):
        mf, n = 1500, 3600
    d = hostlib.synth_decompose_batch(int(rng.integers(0, 1 << 30)), nd, n, mf, 0, mix=decompose_mix)
    hbc = capi.HostBaseCalls([d["signal"][i] for i in range(nd)], [d["bcpos"][i] for i in range(nd)], [d["primary"][i].tobytes() for i in range(nd)],
                             [d["secondary"][i].tobytes() for i in range(nd)])
    # trims / maxindel / MAD cut-off: tracy's defaults in one campaign of two, otherwise values that move the clamps of decomposeAlleles
    dtl, dtr, dmi, dmad = (50, 50, 1000, 5) if args.seed % 2 else [(0, 0, 1000, 5), (13, 77, 300, 9), (30, 30, 40, 0), (50, 50, 7, 5)][(args.seed // 2) % 4]
    got = ctx.decompose_traces([d["profiles"][i] for i in range(nd)], hbc, [d["refs"][i].tobytes() for i in range(nd)], psc, dtl, dtr, dmi, dmad)

    def dwant(i):
        return io.decompose_trace(d["signal"][i], d["bcpos"][i], d["primary"][i].tobytes(), d["secondary"][i].tobytes(), d["refs"][i].tobytes(),
                                  psc, dtl, dtr, dmi, dmad)
    for i, w in enumerate(pool.map(dwant, range(nd))):
        fr = np.asarray(got["fractions"]).reshape(-1, 2)
        if int(got["status"][i]) != w["status"]:
            bad.append(("decompose_status", i, int(got["status"][i]), w["status"]))
            continue
        if w["status"] != 0:  # (the chain refused the trace -- "Alignment of trace to reference failed!": later outputs are unspecified)
            continue
        ok = got["primary"][i] == w["primary"] and got["secdecomp_list"][i] == w["secdecomp"] and (float(fr[i, 0]), float(fr[i, 1])) == w["af"]
        ok = ok and got["dcp"][i] == w["dcp"] and all(got["btr%d" % k][i] == w["btr%d" % k] and int(got["score%d" % k][i]) == w["score%d" % k] for k in range(3))
        if not ok:
            bad.append(("decompose", i))
    done["decompose"] = nd
    ctx.close()
    pool.shutdown()
    return {"compared": done, "mismatches": len(bad), "first": [str(b) for b in bad[:5]], "seed": seed, "lanes": lanes, "pipeline_scoring": list(psc), "decompose_params": [dtl, dtr, dmi, dmad]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1500)
    ap.add_argument("--traces", type=int, default=96)
    ap.add_argument("--lanes", type=int, default=1, help="tracyhip_set_lanes for the pipeline calls")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=1, help="campaigns to run, seeds seed .. seed+rounds-1, lanes alternating 1 / --lanes")
    ap.add_argument("--log", default="", help="append one JSON line per campaign to this file")
    args = ap.parse_args()
    total_bad = 0
    for r in range(args.rounds):
        t0 = time.time()
        res = run_campaign(args.pairs, args.traces, args.lanes if (r % 2 or args.rounds == 1) else 1, args.seed + r)
        res["seconds"] = round(time.time() - t0, 1)
        print(json.dumps(res), flush=True)
        if args.log:
            with open(args.log, "a") as f:
                f.write(json.dumps(res) + "\n")
        total_bad += res["mismatches"]
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
