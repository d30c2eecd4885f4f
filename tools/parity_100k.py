#!/usr/bin/env python
"""parity_100k.py -- the north star's parity sentence run literally: EVERY trace of the two headline batches compared with the
oracle (the CPU restatement of sage.h:191-311 / indigo.h:190-388 under oracle/; test infrastructure, never the product path).

  align      BASELINE.json configs[1]: 10 000 synthetic 1 kb traces vs 10 kb windows (the batch bench.py times, same seeds)
  decompose  BASELINE.json configs[2]: 100 000 synthetic 1 kb heterozygous traces vs 3 kb windows (the batch tools/legs.py times)

Each batch goes through the C ABI (tracyhip_align_traces / tracyhip_decompose_traces, inputs resident in HBM) as ONE call per
mode.  Mode A (both orientation scores exact, one lane) is compared with the oracle trace by trace and field by field:
  align:      score_fwd, score_rev, forward, score_prelim, slice_begin, slice_len, ref_pos, score_final, btr
  decompose:  status, forward, score_fwd, score_rev and -- for traces the chain accepts (status 0) -- breakpoint, score_trim,
              rewritten primary / secondary basecalls, secDecompose, the decomposition table, its verdict, both allelic
              fractions, and per allele: slice geometry, score, btr (three alignments)
The other modes (strand by certificate, two lanes, both) are compared with mode A on the device, array by array (the
certificate's losing-strand score may be its certified upper bound: checked as >= and never deciding differently).  The oracle
consumes the very bytes the GPU consumed (inputs are read back from HBM per block).

One JSON line per block of 1 000 traces (compared, mismatches, first mismatch) -> --out (profiles/r03_parity_100k.jsonl).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCORE = (3, -5, -10, -4)
TRIM = 50
ALIGN_KEYS = ("score_fwd", "score_rev", "forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final")


def usable_cores():
    from bench import usable_cores as uc
    return uc()


class Log:
    def __init__(self, path):
        self.f = open(path, "w") if path else None
        self.lines = []

    def write(self, obj):
        self.lines.append(obj)
        s = json.dumps(obj)
        if self.f:
            self.f.write(s + "\n")
            self.f.flush()
        print(s, flush=True)


# ======================================================================================================================
def run_align(nt, ref_len, trace_len, log, block=1000, seed=1000, dev_index=0, nthreads=None):
    """configs[1]; returns (compared, mismatches)"""
    import pyoracle
    import tracy_amd
    from tracy_amd import capi, hostlib
    pyoracle.lib()
    nthreads = nthreads or usable_cores()
    dev = torch.device("cuda", dev_index)
    n, mf = ref_len, trace_len
    refs, profs, rev = hostlib.synth_align(seed, nt, n, mf, 0)
    d_refs, d_profs = torch.from_numpy(refs).to(dev), torch.from_numpy(profs).to(dev)
    pp_off = np.arange(nt, dtype=np.uint64) * np.uint64(6 * mf)
    pp_len = np.full(nt, mf, dtype=np.uint32)
    rr_off = np.arange(nt, dtype=np.uint64) * np.uint64(n)
    rr_len = np.full(nt, n, dtype=np.uint32)
    ops_cap = mf + n
    ops_off = np.arange(nt, dtype=np.uint64) * np.uint64(ops_cap)
    job = capi.AlignJob()
    job.ntraces = nt
    job.profiles = capi.SeqSet(capi.SEQ_PROFILE, d_profs.data_ptr(), pp_off.ctypes.data_as(C.POINTER(C.c_uint64)), pp_len.ctypes.data_as(C.POINTER(C.c_uint32)), nt)
    job.refs = capi.SeqSet(capi.SEQ_CHAR, d_refs.data_ptr(), rr_off.ctypes.data_as(C.POINTER(C.c_uint64)), rr_len.ctypes.data_as(C.POINTER(C.c_uint32)), nt)
    job.trim_left = job.trim_right = TRIM
    prm = capi.Params(SCORE[0], SCORE[1], SCORE[2], SCORE[3], 1, 0)
    ctx = tracy_amd.Context(dev_index)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lib = capi.lib()

    def one_mode(cert, lanes):
        r = {k: torch.zeros(nt, dtype=torch.int32, device=dev) for k in ALIGN_KEYS if k != "forward"}
        r["ops_len"] = torch.zeros(nt, dtype=torch.int32, device=dev)
        r["forward"] = torch.zeros(nt, dtype=torch.uint8, device=dev)
        r["ops"] = torch.zeros(nt * ops_cap, dtype=torch.uint8, device=dev)
        out = capi.AlignResult()
        for k, v in r.items():
            setattr(out, k, v.data_ptr())
        out.ops_offset = ops_off.ctypes.data_as(C.POINTER(C.c_uint64))
        job.strand_by_certificate = 1 if cert else 0
        ctx.set_lanes(lanes)
        rc = lib.tracyhip_align_traces(ctx._h, C.byref(job), C.byref(prm), capi.MEM_DEVICE, C.byref(out))
        if rc != 0:
            raise RuntimeError("tracyhip_align_traces: %s" % lib.tracyhip_last_error().decode())
        torch.cuda.synchronize()
        return r

    A = one_mode(False, 1)
    # ---- the other modes against mode A, on the device ----
    others = {}
    for name, cert, lanes in (("certificate_1lane", True, 1), ("exact_2lanes", False, 2), ("certificate_2lanes", True, 2)):
        if lanes > 1 and nt < 128:
            continue
        B = one_mode(cert, lanes)
        bad = torch.zeros(nt, dtype=torch.bool, device=dev)
        for k in ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final", "ops_len"):
            bad |= A[k] != B[k]
        bad |= (A["ops"].view(nt, ops_cap) != B["ops"].view(nt, ops_cap)).any(dim=1)
        if cert:  # the winner's score is exact, the loser's may be its certified upper bound
            fw = A["forward"] != 0
            win_a = torch.where(fw, A["score_fwd"], A["score_rev"]); win_b = torch.where(fw, B["score_fwd"], B["score_rev"])
            los_a = torch.where(fw, A["score_rev"], A["score_fwd"]); los_b = torch.where(fw, B["score_rev"], B["score_fwd"])
            bad |= (win_a != win_b) | (los_b < los_a)
        else:
            bad |= (A["score_fwd"] != B["score_fwd"]) | (A["score_rev"] != B["score_rev"])
        nb = int(bad.sum().item())
        others[name] = {"compared_with_mode_A": nt, "mismatches": nb, "first_mismatch": int(torch.nonzero(bad)[0].item()) if nb else None}
        del B
    ctx.set_lanes(1)
    log.write({"workload": "align", "what": "modes vs mode A (exact scores, one lane), every array on the device", "traces": nt, "modes": others})
    H = {k: v.cpu().numpy() for k, v in A.items() if k != "ops"}
    ops = A["ops"].cpu().numpy().reshape(nt, ops_cap)
    compared = mism = 0
    for b0 in range(0, nt, block):
        b1 = min(nt, b0 + block)
        t0 = time.perf_counter()
        want, _ = pyoracle.sage_chain_batch(d_profs[b0:b1].cpu().numpy(), d_refs[b0:b1].cpu().numpy(), SCORE, TRIM, TRIM, nthreads)
        dt = time.perf_counter() - t0
        first, bad = None, 0
        for j, w in enumerate(want):
            i = b0 + j
            diff = [k for k in ALIGN_KEYS if int(H[k][i]) != int(w[k])]
            if ops[i, :int(H["ops_len"][i])].tobytes() != w["btr"]:
                diff.append("btr")
            if int(w["forward"]) != 1 - int(rev[i]):
                diff.append("strand_of_the_synthetic_trace")
            if diff:
                bad += 1
                if first is None:
                    first = {"trace": i, "fields": diff}
        compared += b1 - b0
        mism += bad
        log.write({"workload": "align", "mode": "A: exact scores, one lane, vs the oracle (sage.h:191-311)", "block": [b0, b1], "compared": b1 - b0, "mismatches": bad,
                   "first_mismatch": first, "oracle_s": round(dt, 2), "oracle_threads": nthreads})
    ctx.close()
    mism += sum(v["mismatches"] for v in others.values())
    return compared, mism


# ======================================================================================================================
def run_decompose(nt, ref_len, trace_len, log, block=1000, dev_index=0, nthreads=None, first=0):
    """configs[2]; returns (compared, mismatches, accepted); first: the batch starts at trace `first` of the seeded job"""
    from indigo_oracle import decompose_trace
    from tools.legs import DecomposeLeg
    nthreads = nthreads or usable_cores()
    dev = torch.device("cuda", dev_index)
    leg = DecomposeLeg(nt, ref_len, trace_len, 0, 1, dev, first=first)
    mf, n, cap = trace_len, ref_len, leg.cap

    def snapshot():
        s = {k: v.clone() for k, v in leg.res.items()}
        s["pri"], s["sec"] = leg.t_pri.clone(), leg.t_sec.clone()
        for k in range(3):
            s["ops%d" % k], s["olen%d" % k], s["sc%d" % k] = leg.keep[k][1].clone(), leg.keep[k][2].clone(), leg.keep[k][3].clone()
        return s

    leg.job.strand_by_certificate = 0
    leg.ctx.set_lanes(1)
    leg.step()
    torch.cuda.synchronize()
    A = snapshot()
    others = {}
    for name, cert, lanes in (("certificate_1lane", 1, 1), ("exact_2lanes", 0, 2)):
        if lanes > 1 and nt < 128:
            continue
        leg.job.strand_by_certificate = cert
        leg.ctx.set_lanes(lanes)
        leg.step()
        torch.cuda.synchronize()
        bad = torch.zeros(nt, dtype=torch.bool, device=dev)
        ok = A["status"] == 0
        for k, v in A.items():
            cur = (leg.res[k] if k in leg.res else leg.t_pri if k == "pri" else leg.t_sec if k == "sec" else
                   leg.keep[int(k[-1])][{"ops": 1, "olen": 2, "sc": 3}[k[:-1]]])
            if k in ("score_fwd", "score_rev") and cert:
                continue
            per = (v.view(nt, -1) != cur.view(nt, -1)).any(dim=1)
            bad |= per if k in ("status", "forward", "score_fwd", "score_rev") else (per & ok)  # outputs of rejected traces are unspecified
        if cert:
            fw = A["forward"] != 0
            win_a = torch.where(fw, A["score_fwd"], A["score_rev"]); win_b = torch.where(fw, leg.res["score_fwd"], leg.res["score_rev"])
            los_a = torch.where(fw, A["score_rev"], A["score_fwd"]); los_b = torch.where(fw, leg.res["score_rev"], leg.res["score_fwd"])
            bad |= (win_a != win_b) | (los_b < los_a)
        nb = int(bad.sum().item())
        others[name] = {"compared_with_mode_A": nt, "mismatches": nb, "first_mismatch": int(torch.nonzero(bad)[0].item()) if nb else None}
    leg.ctx.set_lanes(1)
    leg.job.strand_by_certificate = 0
    log.write({"workload": "decompose", "what": "modes vs mode A (exact scores, one lane), every array on the device", "traces": nt, "modes": others})

    H = {k: v.cpu().numpy() for k, v in A.items()}
    bp = H["bp"].reshape(nt, 4)
    dst = H["dstatus"].reshape(nt, 6)
    di, de = H["dcp_indel"].reshape(nt, cap), H["dcp_err"].reshape(nt, cap)
    fr = H["fractions"].reshape(nt, 2)
    pri, sec, sd = H["pri"].reshape(nt, mf), H["sec"].reshape(nt, mf), H["secdecomp"].reshape(nt, mf)
    compared = mism = accepted = 0
    for b0 in range(0, nt, block):
        b1 = min(nt, b0 + block)
        sig = leg.t_sig[b0:b1].cpu().numpy()
        pos = leg.t_pos[b0:b1].cpu().numpy()
        p0 = leg.pri0[b0:b1].cpu().numpy()
        s0 = leg.sec0[b0:b1].cpu().numpy()
        rf = leg.t_ref[b0:b1].cpu().numpy()

        def one(j):
            return decompose_trace(sig[j], pos[j], p0[j].tobytes(), s0[j].tobytes(), rf[j].tobytes(), SCORE)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nthreads) as ex:
            want = list(ex.map(one, range(b1 - b0)))
        dt = time.perf_counter() - t0
        first, bad, acc = None, 0, 0
        for j, w in enumerate(want):
            i = b0 + j
            diff = []
            if int(H["status"][i]) != w["status"]: diff.append("status")
            if int(H["forward"][i]) != w["forward"]: diff.append("forward")
            if int(H["score_fwd"][i]) != w["score_fwd"]: diff.append("score_fwd")
            if int(H["score_rev"][i]) != w["score_rev"]: diff.append("score_rev")
            if w["status"] == 0:
                acc += 1
                wb = w["bp"]
                if (int(bp[i, 0]), int(bp[i, 1]), int(bp[i, 2]) & 0xffffffff) != (int(wb.indelshift), int(wb.traceleft), int(wb.breakpoint)): diff.append("breakpoint")
                if int(H["score_trim"][i]) != w["score_trim"]: diff.append("score_trim")
                if pri[i].tobytes() != w["primary"]: diff.append("primary")
                if sec[i].tobytes() != w["secondary"]: diff.append("secondary")
                if sd[i].tobytes() != w["secdecomp"]: diff.append("secdecomp")
                nd = int(dst[i, 4])
                if [(int(di[i, q]), int(de[i, q])) for q in range(nd)] != [tuple(x) for x in w["dcp"]]: diff.append("dcp")
                if tuple(int(x) for x in dst[i, :4]) != tuple(int(x) for x in w["dstatus"]): diff.append("dstatus")
                if (float(fr[i, 0]), float(fr[i, 1])) != w["af"]: diff.append("allelic_fractions")
                for k in range(3):
                    capk = mf + (n if k < 2 else mf)
                    ln = int(H["olen%d" % k][i])
                    if int(H["sc%d" % k][i]) != w["score%d" % k]: diff.append("score%d" % k)
                    if H["ops%d" % k][i * capk:i * capk + ln].tobytes() != w["btr%d" % k]: diff.append("btr%d" % k)
                    if k < 2:
                        for nm in ("slice_begin", "slice_len", "ref_pos"):
                            if int(H["%s%d" % (nm, k)][i]) != w["%s%d" % (nm, k)]: diff.append("%s%d" % (nm, k))
            if diff:
                bad += 1
                if first is None:
                    first = {"trace": i, "fields": diff}
        compared += b1 - b0
        mism += bad
        accepted += acc
        log.write({"workload": "decompose", "mode": "A: exact scores, one lane, vs the oracle (indigo.h:190-388)", "block": [b0, b1], "compared": b1 - b0,
                   "accepted_by_the_chain": acc, "mismatches": bad, "first_mismatch": first, "oracle_s": round(dt, 2), "oracle_threads": nthreads})
    leg.ctx.close()
    mism += sum(v["mismatches"] for v in others.values())
    return compared, mism, accepted


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--align-traces", type=int, default=10000)
    ap.add_argument("--decompose-traces", type=int, default=100000)
    ap.add_argument("--block", type=int, default=1000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_100k.jsonl"))
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    log = Log(args.out)
    nth = args.threads or usable_cores()
    t0 = time.perf_counter()
    log.write({"what": "parity of the two headline batches against the oracle, every trace (tools/parity_100k.py)", "gpu": torch.cuda.get_device_name(0),
               "oracle_threads": nth, "align_traces": args.align_traces, "decompose_traces": args.decompose_traces})
    tot_c = tot_m = 0
    if args.align_traces:
        c, m = run_align(args.align_traces, 10000, 1000, log, args.block, nthreads=nth)
        tot_c += c; tot_m += m
        log.write({"workload": "align", "summary": True, "compared": c, "mismatches": m})
    if args.decompose_traces:
        c, m, acc = run_decompose(args.decompose_traces, 3000, 1000, log, args.block, nthreads=nth)
        tot_c += c; tot_m += m
        log.write({"workload": "decompose", "summary": True, "compared": c, "accepted_by_the_chain": acc, "mismatches": m})
    log.write({"summary": True, "traces_compared": tot_c, "mismatches": tot_m, "bit_identical": tot_m == 0, "wall_s": round(time.perf_counter() - t0, 1)})
    sys.exit(0 if tot_m == 0 else 1)


if __name__ == "__main__":
    main()
