#!/usr/bin/env python
"""bench_decompose.py -- `tracy decompose` hot section (indigo.h:190-388) on one MI355X (BASELINE.json configs[2]).

A step = tracyhip_decompose_traces over a batch of synthetic heterozygous traces (1 kb trace, 3 kb reference
window: trace + 2*maxindel as produced by k-mer seeding): findBreakpoint, 2 orientation scores, traceback of
the trimmed profile, decomposeAlleles, generateSecondaryDecomposed, allelicFraction, 4 string x string
traceback alignments vs the window / trimmed slices and the global allele1-vs-allele2 alignment.  Inputs are
resident in HBM.  Prints one JSON line (traces/s and GCUPS over the 8 Gotoh calls per trace) plus a CPU
baseline (the oracle's indigo.h chain on a sample) and a bit-exactness check on that sample.
This is a secondary measurement; bench.py (configs[1], `tracy align`) is the contract benchmark.
"""
import argparse
import ctypes as C
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
    ap.add_argument("--traces", type=int, default=20000)
    ap.add_argument("--ref-len", type=int, default=3000)
    ap.add_argument("--trace-len", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=32)
    ap.add_argument("--extra-legs", type=int, default=1, help="also time the strand-certificate and two-lane legs (0: headline leg only, for profiling)")
    args = ap.parse_args()
    import tracy_amd
    from tracy_amd import capi, hostlib
    nt, n, mf = args.traces, args.ref_len, args.trace_len
    d = hostlib.synth_decompose_batch(5000, nt, n, mf, 0)
    ns = d["signal"].shape[2]
    dev = torch.device("cuda", 0)
    t_sig = torch.from_numpy(d["signal"]).to(dev)
    t_pos = torch.from_numpy(d["bcpos"]).to(dev)
    t_prof = torch.from_numpy(d["profiles"]).to(dev)
    t_ref = torch.from_numpy(d["refs"]).to(dev)
    pri0 = torch.from_numpy(d["primary"]).to(dev)
    sec0 = torch.from_numpy(d["secondary"]).to(dev)
    t_pri = pri0.clone()
    t_sec = sec0.clone()

    def u64(a):
        return a.ctypes.data_as(C.POINTER(C.c_uint64))

    def u32(a):
        return a.ctypes.data_as(C.POINTER(C.c_uint32))
    idx = np.arange(nt, dtype=np.uint64)
    sig_off, bc_off, prof_off, ref_off = idx * np.uint64(4 * ns), idx * np.uint64(mf), idx * np.uint64(6 * mf), idx * np.uint64(n)
    nsamp = np.full(nt, ns, np.uint32)
    bc_len = np.full(nt, mf, np.uint32)
    ref_len = np.full(nt, n, np.uint32)
    maxindel = 1000
    cap = 2 * maxindel + 2
    dcp_off = idx * np.uint64(cap)
    job = capi.DecomposeJob()
    job.ntraces = nt
    job.profiles = capi.SeqSet(capi.SEQ_PROFILE, t_prof.data_ptr(), u64(prof_off), u32(bc_len), nt)
    job.bc = capi.BaseCallsBatch(nt, t_sig.data_ptr(), u64(sig_off), u32(nsamp), t_pos.data_ptr(), t_pri.data_ptr(), t_sec.data_ptr(),
                                 u64(bc_off), u32(bc_len))
    job.refs = capi.SeqSet(capi.SEQ_CHAR, t_ref.data_ptr(), u64(ref_off), u32(ref_len), nt)
    job.dprm = capi.DecompParams(50, 50, maxindel, 5)
    job.strand_by_certificate = 0  # headline leg: both orientations swept in full; the certificate mode is timed after it
    res = {
        "bp": torch.zeros(nt * 4, dtype=torch.int32, device=dev), "status": torch.zeros(nt, dtype=torch.int32, device=dev),
        "score_fwd": torch.zeros(nt, dtype=torch.int32, device=dev), "score_rev": torch.zeros(nt, dtype=torch.int32, device=dev),
        "forward": torch.zeros(nt, dtype=torch.uint8, device=dev), "score_trim": torch.zeros(nt, dtype=torch.int32, device=dev),
        "dcp_indel": torch.zeros(nt * cap, dtype=torch.int32, device=dev), "dcp_err": torch.zeros(nt * cap, dtype=torch.int32, device=dev),
        "dstatus": torch.zeros(nt * 6, dtype=torch.int32, device=dev), "secdecomp": torch.zeros(nt * mf, dtype=torch.uint8, device=dev),
        "fractions": torch.zeros(nt * 2, dtype=torch.float64, device=dev),
    }
    out = capi.DecomposeResult()
    for k, v in res.items():
        setattr(out, k, v.data_ptr())
    out.dcp_offset = u64(dcp_off)
    keep = []
    for k in range(3):
        capk = mf + (n if k < 2 else mf)
        off = idx * np.uint64(capk)
        ops = torch.zeros(nt * capk, dtype=torch.uint8, device=dev)
        olen = torch.zeros(nt, dtype=torch.int32, device=dev)
        sc = torch.zeros(nt, dtype=torch.int32, device=dev)
        out.score[k] = sc.data_ptr()
        out.ops[k] = ops.data_ptr()
        out.ops_offset[k] = u64(off)
        out.ops_len[k] = olen.data_ptr()
        keep.append((off, ops, olen, sc, capk))
        if k < 2:
            for nm in ("slice_begin", "slice_len", "ref_pos"):
                a = torch.zeros(nt, dtype=torch.int32, device=dev)
                getattr(out, nm)[k] = a.data_ptr()
                res["%s%d" % (nm, k)] = a
    prm = capi.Params(SCORE[0], SCORE[1], SCORE[2], SCORE[3], 1, 0)
    ctx = tracy_amd.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lib = capi.lib()

    def step():
        t_pri.copy_(pri0)  # decomposeAlleles rewrites the basecalls in place: start every step from the originals
        t_sec.copy_(sec0)
        rc = lib.tracyhip_decompose_traces(ctx._h, C.byref(job), C.byref(prm), capi.MEM_DEVICE, C.byref(out))
        if rc != 0:
            raise RuntimeError(lib.tracyhip_last_error().decode())

    def leg():
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    dt = leg()
    dt_cert = dt_lanes = dt_lanes_cert = 0.0
    same = True
    if args.extra_legs:
        snap = {k: v.clone() for k, v in res.items() if k not in ("score_fwd", "score_rev")}
        snap_ops = [x[1].clone() for x in keep]
        job.strand_by_certificate = 1  # the library's default: strand by certificate
        dt_cert = leg()
        same = all(torch.equal(snap[k], res[k]) for k in snap) and all(torch.equal(a, x[1]) for a, x in zip(snap_ops, keep))
        job.strand_by_certificate = 0
        ctx.set_lanes(2)  # the same batch as two chunks in flight (tracyhip_set_lanes)
        dt_lanes = leg()
        job.strand_by_certificate = 1
        dt_lanes_cert = leg()
        job.strand_by_certificate = 0
        ctx.set_lanes(1)
    mt = mf - 100
    sl = [res["slice_len%d" % k].cpu().numpy().astype(np.int64) for k in range(2)]
    cells = 3 * mt * n * nt + 2 * mt * n * nt + int((mt * sl[0]).sum() + (mt * sl[1]).sum()) + mt * mt * nt
    status = res["status"].cpu().numpy()
    line = {"metric": "traces/s (tracy decompose hot section, indigo.h:190-388)", "value": round(nt * args.steps / dt, 1), "unit": "traces/s",
            "gcups": round(cells * args.steps / dt / 1e9, 1), "ms_per_step": round(dt / args.steps * 1e3, 2), "n_gpus": 1,
            "config": {"workload": "configs[2]: %d synthetic heterozygous %d-base traces `decompose` vs %d-base windows" % (nt, mf, n)},
            "traces_ok": int((status == 0).sum()), "data": "synthetic"}
    if args.extra_legs:
        line["strand_by_certificate"] = {"ms_per_step": round(dt_cert / args.steps * 1e3, 2), "traces_per_s": round(nt * args.steps / dt_cert, 1),
                                         "results_identical_to_headline_leg": bool(same)}
        line["lanes"] = {"lanes": 2, "ms_per_step": round(dt_lanes / args.steps * 1e3, 2), "traces_per_s": round(nt * args.steps / dt_lanes, 1),
                         "strand_by_certificate": {"ms_per_step": round(dt_lanes_cert / args.steps * 1e3, 2),
                                                   "traces_per_s": round(nt * args.steps / dt_lanes_cert, 1)}}
    if args.cpu_sample > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from concurrent.futures import ThreadPoolExecutor
        from indigo_oracle import decompose_trace
        ns_ = min(args.cpu_sample, nt)
        nthreads = min(os.cpu_count() or 1, 16)
        work = list(range(ns_))

        def one(i):
            return decompose_trace(d["signal"][i], d["bcpos"][i], d["primary"][i].tobytes(), d["secondary"][i].tobytes(),
                                   d["refs"][i].tobytes(), SCORE)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nthreads) as ex:
            want = list(ex.map(one, work))
        cdt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(ns_ / cdt, 2), "unit": "traces/s", "cores": min(nthreads, ns_), "kind": "port",
                                "sample": "%d of the same traces through the oracle's indigo.h chain, %.1f s" % (ns_, cdt)}
        pri = t_pri.cpu().numpy().reshape(nt, mf)
        sd = res["secdecomp"].cpu().numpy().reshape(nt, mf)
        fr = res["fractions"].cpu().numpy().reshape(nt, 2)
        ok = True
        for i, w in enumerate(want):
            ok &= pri[i].tobytes() == w["primary"] and sd[i].tobytes() == w["secdecomp"]
            ok &= (float(fr[i, 0]), float(fr[i, 1])) == w["af"]
            for k in range(3):
                off, ops, olen, sc, capk = keep[k]
                ln = int(olen[i].item())
                ok &= int(sc[i].item()) == w["score%d" % k]
                ok &= ops[i * capk:i * capk + ln].cpu().numpy().tobytes() == w["btr%d" % k]
        line["parity_checked"] = {"traces": ns_, "bit_identical": bool(ok)}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
