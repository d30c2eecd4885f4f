"""Workloads bench.py times beside its headline (`tracy align`, configs[1]): BASELINE.json configs[2] (`tracy decompose`,
indigo.h:190-388) and configs[4] (the all-pairs profile x profile scoring of `tracy assemble`, msa.h:33-42).  Each leg
shards its job over the ranks: decompose by contiguous blocks of traces with the final gather in its two halves inside every
timed step (one record per trace, then the three traceback strings, the rewritten basecalls, secDecompose and the decomposition table of
every trace, packed on the device: tracy_amd/shard.py ResultGather); all-pairs by contiguous slices of the upper-triangular pair list with the profiles replicated
and an all_gather of the score slices (the distance matrix ends up on every rank).  No data-path collective.

Only the `cpu_baseline` parts touch oracle/ (the CPU restatement timed on the host cores, and the in-run parity sample)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCORE = (3, -5, -10, -4)
HBM_PEAK_GBS = 8000.0
VALU_PEAK = 78.6       # T lane-ops/s: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
FP32_VECTOR_PEAK = 157.3  # TFLOP/s (MI355X_MICROARCH.md), FMA = 2 flops; separately rounded mul / add reach half of it per issue slot

TIMERS = (("score", 0), ("trace", 1), ("walk", 2), ("band", 3), ("prefix", 4), ("origin", 5), ("decompose", 6), ("allelic_fraction", 7), ("misc", 8), ("front", 9))


def rank_threads(world=None):
    """host threads of this rank: the cores the process may use, divided by the ranks that share the node (every rank of a
    one-node job runs its host stages -- synthesis, k-mer seeding, the CPU baseline -- at the same time)"""
    from bench import usable_cores
    local = int(os.environ.get("LOCAL_WORLD_SIZE", world if world else os.environ.get("WORLD_SIZE", "1")))
    return max(1, usable_cores() // max(1, local))


def u64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def read_timers(lib, ctx):
    from tracy_amd import capi
    kt = capi.KernelTiming()
    res = {}
    for name, which in TIMERS:
        lib.tracyhip_timing_get(ctx._h, which, C.byref(kt))
        res[name] = dict(ms=kt.ms, launches=int(kt.launches), cells=int(kt.cells), bytes=int(kt.bytes))
    return res


def timed(step, steps, warmup, dist, lib=None, ctx=None, after_warmup=None):
    """W untimed + K timed steps between barrier + synchronize; returns (seconds, kernel timers of the timed steps)"""
    for _ in range(warmup):
        step()
    if after_warmup is not None:
        after_warmup()
    if lib is not None:
        lib.tracyhip_timing_enable(ctx._h, 0 if os.environ.get("TRACYHIP_BENCH_NO_TIMERS") else 1)
        lib.tracyhip_timing_reset(ctx._h)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    timers = None
    if lib is not None:
        lib.tracyhip_timing_enable(ctx._h, 0)
        timers = read_timers(lib, ctx)
    return dt, timers


def max_over_ranks(dist, dev, values):
    t = torch.tensor(values, dtype=torch.float64, device="cpu" if (dist is not None and dist.get_backend() == "gloo") else dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def sum_over_ranks(dist, dev, values):
    t = torch.tensor(values, dtype=torch.float64, device="cpu" if (dist is not None and dist.get_backend() == "gloo") else dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t]


def min_over_ranks(dist, dev, values):
    t = torch.tensor(values, dtype=torch.float64, device="cpu" if (dist is not None and dist.get_backend() == "gloo") else dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return [float(x) for x in t]


def pmc_traffic(patterns, tag_glob="r[0-9][0-9]_pmc_hbm*.json"):
    """HBM bytes PER STEP of the kernels whose names match one of `patterns` (regular expressions: the exact instantiations a timer
    covers), from the latest committed rocprofv3 PMC summary (WRITE_SIZE + FETCH_SIZE, separate passes): every matching kernel's
    bytes per launch x its launches, divided by the steps of the profiled run (= the launches of encode_codes_kernel, once per step)"""
    import glob
    import json
    import re
    if isinstance(patterns, str):
        patterns = [patterns]
    rx = [re.compile(p) for p in patterns]
    try:
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", tag_glob)), reverse=True):
            pmc = json.load(open(f))
            steps = max([r["launches"] for r in pmc if "encode_codes_kernel" in r["kernel"]] or [0])
            if not steps:
                steps = min([r["launches"] for r in pmc] or [0])
            tot, seen = 0.0, set()
            for c in ("WRITE_SIZE", "FETCH_SIZE"):
                for r in pmc:
                    if r["counter"] == c and any(x.search(r["kernel"]) for x in rx):
                        tot += r["bytes"] * r["launches"]
                        seen.add(c)
            if steps and len(seen) == 2:
                return int(tot / steps), "profiles/%s (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate passes; per step, summed over the instantiations the timer covers)" % os.path.basename(f)
    except (OSError, ValueError, KeyError, IndexError):
        pass
    return None, None


def pmc_clock(pattern, grid_min=0, tag_glob="r[0-9][0-9]_pmc_stalls.json"):
    """effective shader clock (GHz) under the kernel whose name matches `pattern` (the largest grid at least `grid_min`), from the latest
    committed rocprofv3 pass of GRBM_GUI_ACTIVE (summed over the eight XCDs) / the launch's duration (tools/profile_summarise.py)"""
    import glob
    import json
    import re
    rx = re.compile(pattern)
    try:
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", tag_glob)), reverse=True):
            rows = [r for r in json.load(open(f)) if r["counter"] == "GRBM_GUI_ACTIVE" and rx.search(r["kernel"]) and (r.get("grid") or 0) >= grid_min
                    and r.get("effective_clock_ghz") and (r.get("avg_launch_ms") or 0) > 1.0]
            if rows:
                r = max(rows, key=lambda x: x.get("grid") or 0)
                return float(r["effective_clock_ghz"]), "profiles/%s (GRBM_GUI_ACTIVE / 8 XCDs / launch duration, kernel on its own, grid %s)" % (os.path.basename(f), r.get("grid"))
    except (OSError, ValueError, KeyError, IndexError):
        pass
    return None, None


def kernel_block(name, kernel, t, steps, ops_per_cell=None, flops_per_cell=None, traffic_key=None, traffic_glob="r[0-9][0-9]_pmc_hbm*.json"):
    """roofline object of one kernel class from the library's HIP-event timers.  `bound` names the ceiling the numbers show: integer DP
    with its inputs resident is VALU-issue bound (achieved / peak / frac are then lane-ops per second); the HBM figures stay beside it."""
    ms = t["ms"]
    gbs = t["bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    gc = t["cells"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    per_step, src = pmc_traffic(traffic_key, traffic_glob) if traffic_key else (None, None)  # (the PMC summary of THIS workload: kernels are shared between the legs)
    launches_per_step = max(t["launches"], 1) / max(steps, 1)
    traffic = int(per_step / launches_per_step) if per_step is not None else None
    hbm = {"achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
           "algorithmic_bytes_per_launch": t["bytes"] // max(t["launches"], 1), "traffic": traffic, "traffic_source": src}
    out = {"bound": "hbm", "kernel": kernel, "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac"],
           "traffic": traffic, "traffic_source": src, "avg_launch_ms": round(ms / max(t["launches"], 1), 3), "launches": t["launches"],
           "algorithmic_bytes_per_launch": hbm["algorithmic_bytes_per_launch"], "kernel_gcups": round(gc, 1), "timer": name}
    if ops_per_cell:
        out["valu"] = {"achieved": round(gc * ops_per_cell / 1e3, 2), "peak": VALU_PEAK, "unit": "T lane-ops/s",
                       "frac": round(gc * ops_per_cell / 1e3 / VALU_PEAK, 3), "ops_per_cell": ops_per_cell}
        if out["valu"]["frac"] > hbm["frac"]:  # the ceiling the numbers show
            out.update({"bound": "valu", "achieved": out["valu"]["achieved"], "peak": VALU_PEAK, "unit": "T lane-ops/s", "frac": out["valu"]["frac"], "hbm": hbm})
    if flops_per_cell:
        out["fp32_vector"] = {"achieved": round(gc * flops_per_cell / 1e3, 2), "peak": FP32_VECTOR_PEAK, "unit": "TFLOP/s",
                              "frac": round(gc * flops_per_cell / 1e3 / FP32_VECTOR_PEAK, 3), "flops_per_cell": flops_per_cell,
                              "note": "every product and sum is rounded separately (align.h:112-116): no FMA, so an issue slot carries half of the FMA-counted peak"}
    return out


# =====================================================================================================================
class DecomposeLeg:
    """configs[2]: `total` synthetic 1 kb traces `decompose` vs 3 kb windows (80 % het indel + het SNVs, 10 % homozygous indel only,
    10 % no variant, both strands: SURVEY.md 8d), sharded over the ranks by contiguous blocks."""

    def __init__(self, total, ref_len, trace_len, rank, world, dev, lanes=1, first=0):
        """first: index of the batch's first trace in the seeded sequence (trace i of the 100 000-trace job has seed 5000 + i):
        a parity test compares blocks taken from all over the job"""
        import tracy_amd
        from tracy_amd import capi, hostlib
        from tracy_amd.shard import shard_range
        self.capi, self.total, self.n, self.mf, self.rank, self.world, self.dev = capi, total, ref_len, trace_len, rank, world, dev
        self.lo, self.hi = shard_range(total, rank, world)
        nt = self.nt = self.hi - self.lo
        n, mf = ref_len, trace_len
        t0 = time.perf_counter()
        d = hostlib.synth_decompose_batch(5000 + first + self.lo, nt, n, mf, rank_threads(world), mix=1)
        self.synth_s = time.perf_counter() - t0
        ns = d["signal"].shape[2]
        keep_host = min(nt, 256)  # the CPU baseline / parity sample reads the first traces from the host copy
        self.host = {k: np.ascontiguousarray(v[:keep_host]) for k, v in d.items()}
        self.t_sig = torch.from_numpy(d["signal"]).to(dev)
        self.t_pos = torch.from_numpy(d["bcpos"]).to(dev)
        self.t_prof = torch.from_numpy(d["profiles"]).to(dev)
        self.t_ref = torch.from_numpy(d["refs"]).to(dev)
        self.pri0 = torch.from_numpy(d["primary"]).to(dev)
        self.sec0 = torch.from_numpy(d["secondary"]).to(dev)
        del d
        self.t_pri, self.t_sec = self.pri0.clone(), self.sec0.clone()
        idx = np.arange(nt, dtype=np.uint64)
        self.arrs = dict(sig_off=idx * np.uint64(4 * ns), bc_off=idx * np.uint64(mf), prof_off=idx * np.uint64(6 * mf), ref_off=idx * np.uint64(n),
                         nsamp=np.full(nt, ns, np.uint32), bc_len=np.full(nt, mf, np.uint32), ref_len=np.full(nt, n, np.uint32))
        a = self.arrs
        maxindel = 1000
        cap = self.cap = 2 * maxindel + 2
        a["dcp_off"] = idx * np.uint64(cap)
        job = self.job = capi.DecomposeJob()
        job.ntraces = nt
        job.profiles = capi.SeqSet(capi.SEQ_PROFILE, self.t_prof.data_ptr(), u64(a["prof_off"]), u32(a["bc_len"]), nt)
        job.bc = capi.BaseCallsBatch(nt, self.t_sig.data_ptr(), u64(a["sig_off"]), u32(a["nsamp"]), self.t_pos.data_ptr(), self.t_pri.data_ptr(),
                                     self.t_sec.data_ptr(), u64(a["bc_off"]), u32(a["bc_len"]))
        job.refs = capi.SeqSet(capi.SEQ_CHAR, self.t_ref.data_ptr(), u64(a["ref_off"]), u32(a["ref_len"]), nt)
        job.dprm = capi.DecompParams(50, 50, maxindel, 5)
        job.strand_by_certificate = 0  # headline: both orientation scores exact
        z = lambda k, dt: torch.zeros(k, dtype=dt, device=dev)  # noqa: E731
        res = self.res = {"bp": z(nt * 4, torch.int32), "status": z(nt, torch.int32), "score_fwd": z(nt, torch.int32), "score_rev": z(nt, torch.int32),
                          "forward": z(nt, torch.uint8), "score_trim": z(nt, torch.int32), "dcp_indel": z(nt * cap, torch.int32),
                          "dcp_err": z(nt * cap, torch.int32), "dstatus": z(nt * 6, torch.int32), "secdecomp": z(nt * mf, torch.uint8),
                          "fractions": z(nt * 2, torch.float64)}
        out = self.out = capi.DecomposeResult()
        for k, v in res.items():
            setattr(out, k, v.data_ptr())
        out.dcp_offset = u64(a["dcp_off"])
        self.keep = []
        for k in range(3):
            capk = mf + (n if k < 2 else mf)
            off = idx * np.uint64(capk)
            ops, olen, sc = z(nt * capk, torch.uint8), z(nt, torch.int32), z(nt, torch.int32)
            out.score[k], out.ops[k], out.ops_offset[k], out.ops_len[k] = sc.data_ptr(), ops.data_ptr(), u64(off), olen.data_ptr()
            self.keep.append((off, ops, olen, sc, capk))
            if k < 2:
                for nm in ("slice_begin", "slice_len", "ref_pos"):
                    t = z(nt, torch.int32)
                    getattr(out, nm)[k] = t.data_ptr()
                    res["%s%d" % (nm, k)] = t
        self.prm = capi.Params(SCORE[0], SCORE[1], SCORE[2], SCORE[3], 1, 0)
        self.ctx = tracy_amd.Context(dev.index or 0)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.ctx.set_lanes(max(1, lanes))
        self.lib = capi.lib()
        self.gatherer, self.gathered, self.gathered_bytes = None, None, 0
        self.gather_seconds, self.gather_steps = 0.0, 0
        self.bc_len_col = torch.full((nt, 1), mf, dtype=torch.int32, device=dev)

    def step(self, dist=None):
        self.t_pri.copy_(self.pri0)  # decomposeAlleles rewrites the basecalls in place: start every step from the originals
        self.t_sec.copy_(self.sec0)
        rc = self.lib.tracyhip_decompose_traces(self.ctx._h, C.byref(self.job), C.byref(self.prm), self.capi.MEM_DEVICE, C.byref(self.out))
        if rc != 0:
            raise RuntimeError("tracyhip_decompose_traces: %s" % self.lib.tracyhip_last_error().decode())
        t_g = time.perf_counter()
        if dist is not None:
            # the final gather, both halves: one fixed-size record per trace in one collective, then what has no fixed size --
            # the three traceback strings, the rewritten basecalls, bc.secDecompose, the decomposition table -- packed on the device and
            # shipped in one exchange sized from the records' length columns
            if self.gatherer is None or self.gatherer.dist is not dist:
                from tracy_amd.shard import ResultGather, shard_range
                self.gatherer = ResultGather(dist, [b - a for a, b in (shard_range(self.total, r_, self.world) for r_ in range(self.world))], self.ctx)
            rec, pay = self.result_records()
            self.gathered = self.gatherer.gather(rec, pay)
            self.gathered_bytes = self.gatherer.bytes_last
            self.gather_seconds += time.perf_counter() - t_g
            self.gather_steps += 1

    REC_COLS = ("status", "score_trim", "score_fwd", "score_rev", "forward", "score0", "score1", "score2", "ops_len0", "ops_len1", "ops_len2", "slice_begin0",
                "slice_len0", "ref_pos0", "slice_begin1", "slice_len1", "ref_pos1", "bc_len", "bp.indelshift", "bp.traceleft", "bp.breakpoint", "bp.best_diff",
                "ds.kind", "ds.best_ins", "ds.best_del", "ds.best_fr", "ds.dcp_n", "ds.pad", "af0.lo", "af0.hi", "af1.lo", "af1.hi")

    def result_records(self):
        """this rank's results as (int32 record per trace, ragged payloads [(buffer, stride in elements, record column of its length)])"""
        r, nt, mf = self.res, self.nt, self.mf
        n = self.job.ntraces
        col = lambda t: t[:n].reshape(n, 1)  # noqa: E731
        parts = [col(r["status"]), col(r["score_trim"]), col(r["score_fwd"]), col(r["score_rev"]), col(r["forward"]).to(torch.int32)]
        parts += [col(self.keep[k][3]) for k in range(3)] + [col(self.keep[k][2]) for k in range(3)]
        parts += [col(r["%s%d" % (nm, k)]) for k in range(2) for nm in ("slice_begin", "slice_len", "ref_pos")]
        parts += [self.bc_len_col[:n], r["bp"].view(nt, 4)[:n], r["dstatus"].view(nt, 6)[:n], r["fractions"].view(torch.int32).view(nt, 4)[:n]]
        rec = torch.cat(parts, dim=1).contiguous()
        C_ = self.REC_COLS.index
        pay = [(self.keep[k][1], self.keep[k][4], C_("ops_len%d" % k)) for k in range(3)]
        pay += [(self.t_pri, mf, C_("bc_len")), (self.t_sec, mf, C_("bc_len")), (r["secdecomp"], mf, C_("bc_len")),
                (r["dcp_indel"], self.cap, C_("ds.dcp_n")), (r["dcp_err"], self.cap, C_("ds.dcp_n"))]
        return rec, pay

    def cells(self):
        mt = self.mf - 100
        sl = [self.res["slice_len%d" % k].cpu().numpy().astype(np.int64) for k in range(2)]
        # 3 profile DPs (2 orientation scores + the traceback of the trimmed trace), 2 x gotoh(allele, window), 2 x gotoh(allele, slice), pri vs sec
        return int(3 * mt * self.n * self.nt + 2 * mt * self.n * self.nt + (mt * sl[0]).sum() + (mt * sl[1]).sum() + mt * mt * self.nt)

    def small_batch(self, small, steps=6):
        """the first `small` traces of this rank's block as a call of their own (the shard an 8-GPU job of 100 000 traces gives every
        rank): what a step costs when the kernels no longer fill the device for long -- fixed latencies do not shrink with the batch"""
        nt = self.nt
        if small <= 0 or small >= nt:
            return None
        self.job.ntraces = small
        self.job.bc.ntraces = small
        from tracy_amd.shard import ResultGather

        class _Solo:  # (the shard has no peer on this box: records and payloads are built and packed as its rank would, nothing is sent)
            def __init__(self, ctx):
                self.g = ResultGather(None, [small], ctx)

            def pack(self, rec, pay):
                return int(self.g._pack("solo", rec, pay)[0].numel())
        solo = _Solo(self.ctx)

        def shard_step():
            self.step(None)
            return solo.pack(*self.result_records())
        try:
            for _ in range(2):
                shard_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                packed_bytes = shard_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(None)
            torch.cuda.synchronize()
            dt_call = (time.perf_counter() - t0) / steps
            st = self.ctx.last_call_stats()
            # the same call cut into two chunks on two streams (tracyhip_set_lanes): one chunk's chain of short launches beside the other's
            self.ctx.set_lanes(2)
            for _ in range(2):
                self.step(None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(None)
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / steps
        finally:
            self.ctx.set_lanes(1)
            self.job.ntraces = nt
            self.job.bc.ntraces = nt
        return {"traces": small, "ms_per_step": round(dt * 1e3, 3), "ms_per_call_without_packing": round(dt_call * 1e3, 3), "packed_bytes": int(packed_bytes),
                "traces_per_s": round(small / dt, 1), "steps": steps,
                "stream_ordered": st["stream_ordered"], "host_syncs": st["host_syncs"], "fallback_traces": st["fallback_traces"],
                "two_lanes_ms_per_step": round(dt2 * 1e3, 3)}

    def run(self, dist, steps, warmup, extra_legs=True, cpu_sample=64):
        dev = self.dev
        def reset_gather_clock():  # (the first gather sets the communicator up)
            self.gather_seconds, self.gather_steps = 0.0, 0
        dt, timers = timed(lambda: self.step(dist), steps, warmup, dist, self.lib, self.ctx, after_warmup=reset_gather_clock)
        gather_ms = self.gather_seconds / max(self.gather_steps, 1) * 1e3
        call_stats = self.ctx.last_call_stats()
        dt_rank = dt
        small = self.small_batch(max(1, self.total // 8)) if (self.world == 1 and extra_legs) else None
        cells = self.cells()
        ok_traces = int((self.res["status"] == 0).sum().item())
        snap = {k: v.clone() for k, v in self.res.items() if k not in ("score_fwd", "score_rev")}
        snap_ops = [x[1].clone() for x in self.keep]
        snap_pri = self.t_pri.clone()
        dt_cert = dt_lanes = 0.0
        same = True
        if extra_legs:
            self.job.strand_by_certificate = 1
            dt_cert, _ = timed(lambda: self.step(dist), steps, warmup, dist)
            same = all(torch.equal(snap[k], self.res[k]) for k in snap) and all(torch.equal(a, x[1]) for a, x in zip(snap_ops, self.keep))
            self.job.strand_by_certificate = 0
            self.ctx.set_lanes(2)
            dt_lanes, _ = timed(lambda: self.step(dist), steps, warmup, dist)
            same = same and all(torch.equal(snap[k], self.res[k]) for k in snap)
            self.job.strand_by_certificate = 1
            dt_both, _ = timed(lambda: self.step(dist), steps, warmup, dist)
            same = same and all(torch.equal(snap[k], self.res[k]) for k in snap)
            self.job.strand_by_certificate = 0
            self.ctx.set_lanes(1)
        else:
            dt_both = 0.0
        dt, dt_cert, dt_lanes, dt_both = max_over_ranks(dist, dev, [dt, dt_cert, dt_lanes, dt_both])
        (dt_min,) = min_over_ranks(dist, dev, [dt_rank])
        cells_all, ok_all, nt_all, gbytes_all = sum_over_ranks(dist, dev, [float(cells), float(ok_traces), float(self.nt), float(self.gathered_bytes)])
        gather_ok = None
        if self.rank == 0 and self.gathered is not None and self.gathered[0] is not None:
            self.step(dist)  # (the extra legs above ran other modes: one more headline step, whose gather is checked)
            rec, pay = self.result_records()
            gather_ok = self.gatherer.check_own_block(self.gathered[0], self.gathered[1], rec, pay)
        elif dist is not None and self.world > 1:
            self.step(dist)
        self.gathered = None
        if self.rank != 0:
            return None
        tot_ms = sum(timers[k]["ms"] for k, _ in TIMERS)
        dom = max((k for k, _ in TIMERS), key=lambda k: timers[k]["ms"])
        names = {"score": ("gotoh_ckpt_prefix_kernel<K,16,compact,8> + gotoh_prefix_kernel<8,16,compact,strings> (16-bit sweeps: the strand the k-mer vote "
                           "does not pick in full, row m kept; the 128-row prefixes of the voted strand and of both alleles over the whole window, row 128 "
                           "kept for the band below it; cells credited: the rows swept)", 8.0, [r"gotoh_ckpt_prefix_kernel<", r"gotoh_prefix_kernel<"]),
                 "front": ("front_place + band16_cont16_kernel<K> + front_certify (pruned sweeps: the rows below the prefix on the diagonals around its best "
                           "column on 16-bit cells, certified per pair)", 9.0, [r"band16_cont(16)?_kernel<", r"front_place_kernel", r"front_certify_kernel"]),
                 "origin": ("band16_kernel<K,1> (gotoh(allele, window) whose alignment only trimReferenceSlice reads: origin-tracking sweep on the band its score allows)", 11.0, [r"band16_kernel<\d+, 1>", r"band16_multi3?(_counted)?_kernel<1>"]),
                 "trace": ("band16_kernel<K,0> (tracebacks on diagonal bands, four pairs per wave: trimmed trace vs window, allele vs trimmed slice, allele 1 vs allele 2; cells / bytes: the bands')", 14.0, [r"band16_kernel<\d+, 0>", r"band16_multi3?(_counted)?_kernel<0>"]),
                 "band": ("gotoh_band_kernel<K,QP> (band traceback of the trimmed trace)", 14.0, "gotoh_band_kernel"),
                 "walk": ("gotoh_walk_kernel", None, "gotoh_walk_kernel"), "prefix": ("gotoh_prefix_kernel", 8.0, "gotoh_prefix_kernel"),
                 "decompose": ("decompose_kernel (decomposeAlleles, decompose.h:179-376)", None, "decompose_kernel"),
                 "allelic_fraction": ("allelic_fraction_kernel (decompose.h:412-621)", None, "allelic_fraction_kernel"),
                 "misc": ("breakpoint / homozygous / secdecomp kernels", None, None)}
        dec_glob = "r[0-9][0-9]_pmc_hbm_decompose.json"
        roof = kernel_block(dom, names[dom][0], timers[dom], steps, ops_per_cell=names[dom][1], traffic_key=names[dom][2], traffic_glob=dec_glob)
        roof["share_of_kernel_time"] = round(timers[dom]["ms"] / tot_ms, 3) if tot_ms else None
        roof["ms_per_step"] = {k: round(timers[k]["ms"] / steps, 3) for k, _ in TIMERS if timers[k]["ms"] > 0}
        roof["ms_per_step_note"] = ("event-to-event times of the kernel classes; with side streams on (tracyhip option no_fork = 0) allelic_fraction runs beside "
                                    "the allele stages and the voted strand's chain (front) beside the sweeps (score), so these intervals overlap and their sum "
                                    "exceeds the step; the step itself is `ms_per_step` of the line")
        roof["other_kernels"] = {k: {kk: vv for kk, vv in kernel_block(k, names[k][0], timers[k], steps, ops_per_cell=names[k][1], traffic_key=names[k][2], traffic_glob=dec_glob).items()
                                     if kk in ("kernel", "achieved", "frac", "traffic", "avg_launch_ms", "kernel_gcups", "valu", "algorithmic_bytes_per_launch")}
                                 for k in ("score", "front", "origin", "trace") if k in timers and k != dom and timers[k]["ms"] > 0}
        line = {"metric": "traces/s (tracy decompose hot section, indigo.h:190-388)", "value": round(nt_all * steps / dt, 1), "unit": "traces/s",
                "gcups": round(cells_all * steps / dt / 1e9, 1), "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps, "warmup": warmup,
                "n_gpus": self.world, "scaling": "strong", "dtype": "int16 (orientation sweeps) / int32 (tracebacks, origin sweeps) / f64 (allelicFraction)",
                "config": {"workload": "configs[2]: %d synthetic %d-base traces `decompose` vs %d-base windows (80%% het indel + het SNVs, 10%% homozygous "
                                       "indel, 10%% no variant, both strands), sharded over %d rank(s)" % (int(nt_all), self.mf, self.n, self.world),
                           "traces_total": int(nt_all), "trace_len": self.mf, "ref_len": self.n},
                "traces_ok": int(ok_all), "gathered_bytes_per_step": int(gbytes_all), "gather_checked": gather_ok, "gather_ms_per_step": round(gather_ms, 3), "data": "synthetic", "synthesis_s": round(self.synth_s, 1), "roofline": roof,
                # one rank's view of the call: planned on the device, one host synchronisation (stream.hip); min / max over the ranks of a sharded job
                "pipeline": {"stream_ordered": call_stats["stream_ordered"], "host_syncs_per_call": call_stats["host_syncs"],
                             "traces_to_host_planned_tiers": call_stats["fallback_traces"], "traces_per_rank": self.nt,
                             "ms_per_step_rank_min": round(dt_min / steps * 1e3, 2), "ms_per_step_rank_max": round(dt / steps * 1e3, 2),
                             "host_threads_per_rank": rank_threads(self.world)}}
        if small is not None:
            small["vs_eighth_of_the_full_step"] = round(small["ms_per_step"] / (dt / steps * 1e3 / 8.0), 3)
            small["note"] = "the shard one of 8 GPUs gets from this job, as a call of its own on this GPU: strong scaling stays near-linear while this ratio stays near 1"
            line["small_batch"] = small
        if extra_legs:
            line["strand_by_certificate"] = {"ms_per_step": round(dt_cert / steps * 1e3, 2), "traces_per_s": round(nt_all * steps / dt_cert, 1),
                                             "results_identical_to_headline_leg": bool(same)}
            line["lanes"] = {"lanes": 2, "ms_per_step": round(dt_lanes / steps * 1e3, 2), "traces_per_s": round(nt_all * steps / dt_lanes, 1),
                             "strand_by_certificate": {"ms_per_step": round(dt_both / steps * 1e3, 2), "traces_per_s": round(nt_all * steps / dt_both, 1)}}
        if cpu_sample > 0:  # (rank 0; at N > 1 on its share of the host cores)
            line.update(self.cpu_baseline(cpu_sample, snap, snap_ops, snap_pri))
        return line

    def cpu_baseline(self, sample, snap, snap_ops, snap_pri):
        """the oracle's indigo.h chain on the first `sample` traces, one trace per thread; the same traces must be bit-identical on the GPU"""
        for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        from concurrent.futures import ThreadPoolExecutor
        from indigo_oracle import decompose_trace
        from bench import usable_cores
        h = self.host
        ns_ = min(sample, h["signal"].shape[0])
        nthreads = rank_threads(self.world)

        def one(i):
            return decompose_trace(h["signal"][i], h["bcpos"][i], h["primary"][i].tobytes(), h["secondary"][i].tobytes(), h["refs"][i].tobytes(), SCORE)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nthreads) as ex:
            want = list(ex.map(one, range(ns_)))
        cdt = time.perf_counter() - t0
        t1 = time.perf_counter()
        one(0)
        one(1)
        c1 = (time.perf_counter() - t1) / 2
        mf = self.mf
        pri = snap_pri.cpu().numpy().reshape(self.nt, mf)
        sd = snap["secdecomp"].cpu().numpy().reshape(self.nt, mf)
        fr = snap["fractions"].cpu().numpy().reshape(self.nt, 2)
        st = snap["status"].cpu().numpy()
        ok = True
        for i, w in enumerate(want):
            ok &= int(st[i]) == w["status"]
            if w["status"] != 0:
                continue
            ok &= pri[i].tobytes() == w["primary"] and sd[i].tobytes() == w["secdecomp"] and (float(fr[i, 0]), float(fr[i, 1])) == w["af"]
            for k in range(3):
                off, _, olen, sc, capk = self.keep[k]
                ln = int(olen[i].item())
                ok &= int(sc[i].item()) == w["score%d" % k]
                ok &= snap_ops[k][i * capk:i * capk + ln].cpu().numpy().tobytes() == w["btr%d" % k]
        return {"cpu_baseline": {"value": round(ns_ / cdt, 2), "unit": "traces/s", "cores": min(nthreads, ns_), "kind": "port",
                                 "sample": "%d of the same traces through the oracle's indigo.h chain (C via ctypes, one trace per thread), %.1f s" % (ns_, cdt),
                                 "single_thread": {"value": round(1.0 / c1, 3), "unit": "traces/s"}},
                "parity_checked": {"traces": ns_, "bit_identical": bool(ok)}}


# =====================================================================================================================
class AllPairsLeg:
    """configs[4]: all-pairs profile x profile gotohScore<true,true> over `ntr` overlapping ~1 kb traces (msa.h:33-42 distanceMatrix).
    The upper-triangular pair list is cut into contiguous slices of equal cell count, one per rank; the profiles (24 KB each) are
    replicated; the score slices are all-gathered, so every rank ends with the whole matrix."""

    def __init__(self, ntr, trace_len, rank, world, dev):
        import tracy_amd
        from tracy_amd import capi, hostlib
        from tracy_amd.shard import pair_slice
        self.capi, self.ntr, self.mf, self.rank, self.world, self.dev = capi, ntr, trace_len, rank, world, dev
        # traces tiled over one region, trace i starting at i * step (neighbours overlap; the DP cost does not depend on it)
        refs, profs, rev = hostlib.synth_align(9000, ntr, 2 * trace_len + 200, trace_len, rank_threads(world))
        self.profs = np.ascontiguousarray(profs)
        self.lens = np.full(ntr, trace_len, np.uint32)
        i1, i2, self.bounds = pair_slice(self.lens, rank, world)
        self.npairs = int(self.bounds[-1])
        self.i1, self.i2 = np.ascontiguousarray(i1, np.uint32), np.ascontiguousarray(i2, np.uint32)
        self.t_prof = torch.from_numpy(self.profs).to(dev)
        self.off = np.arange(ntr, dtype=np.uint64) * np.uint64(6 * trace_len)
        pr = self.pairs = capi.Pairs()
        pr.npairs = len(self.i1)
        pr.a1 = capi.SeqSet(capi.SEQ_PROFILE, self.t_prof.data_ptr(), u64(self.off), u32(self.lens), ntr)
        pr.a2 = capi.SeqSet(capi.SEQ_PROFILE, self.t_prof.data_ptr(), u64(self.off), u32(self.lens), ntr)
        pr.a1_index, pr.a2_index = u32(self.i1), u32(self.i2)
        self.scores = torch.zeros(max(len(self.i1), 1), dtype=torch.int32, device=dev)
        self.prm = capi.Params(SCORE[0], SCORE[1], SCORE[2], SCORE[3], 1, 1)
        self.ctx = tracy_amd.Context(dev.index or 0)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.lib = capi.lib()
        self.matrix = None

    def step(self, dist=None):
        rc = self.lib.tracyhip_gotoh_score(self.ctx._h, C.byref(self.pairs), C.byref(self.prm), self.capi.MEM_DEVICE, C.c_void_p(self.scores.data_ptr()))
        if rc != 0:
            raise RuntimeError("tracyhip_gotoh_score: %s" % self.lib.tracyhip_last_error().decode())
        if dist is not None:
            from tracy_amd.shard import all_gather_slices
            self.matrix = all_gather_slices(dist, self.scores[:len(self.i1)], self.bounds)
        else:
            self.matrix = self.scores[:len(self.i1)]

    def run(self, dist, steps, warmup, cpu_sample=4096):
        dev = self.dev
        dt, timers = timed(lambda: self.step(dist), steps, warmup, dist, self.lib, self.ctx)
        (dt,) = max_over_ranks(dist, dev, [dt])
        if self.rank != 0:
            return None
        mf = self.mf
        cells = float(self.npairs) * mf * mf
        # 16-term body (row N zero in both profiles): per cell 16 x (mul, mul, add) fp32 + the Gotoh cell
        # 23.8 VALU lane-ops per reference cell: rocprofv3 SQ_INSTS_VALU x 64 / cells of the launch (profiles/rNN_pmc_valu_allpairs.json)
        roof = kernel_block("score", "gotoh_prof_kernel<8,score,NT=4> (profile x profile Gotoh cell; substitution score = the int of the 16-term "
                            "fp32 chain, taken from a screened 4-fma short form where a proven margin excludes every integer, from per-row "
                            "tables against one-hot / uniform columns, from the chain itself otherwise; 25-term twin for profiles with N weight)",
                            timers["score"], steps, ops_per_cell=23.8, traffic_key="gotoh_prof_kernel", traffic_glob="r[0-9][0-9]_pmc_hbm_allpairs.json")
        line = {"metric": "GCUPS (all-pairs profile x profile gotohScore<true,true>, msa.h:33-42)", "value": round(cells * steps / dt / 1e9, 1), "unit": "GCUPS",
                "pairs": int(self.npairs), "pairs_per_s": round(self.npairs * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps,
                "warmup": warmup, "n_gpus": self.world, "scaling": "strong", "dtype": "f32 (substitution scores: the ints of align.h:112-117) / int32 (DP)",
                "config": {"workload": "configs[4]: %d traces of %d bases, %d pairs, pair list sharded over %d rank(s), profiles replicated, "
                                       "score slices all-gathered" % (self.ntr, mf, self.npairs, self.world), "traces": self.ntr, "trace_len": mf},
                "data": "synthetic (profiles resident in HBM, index arrays on the host as the ABI defines)", "roofline": roof}
        if cpu_sample > 0:  # (rank 0; at N > 1 on its share of the host cores and of the pair list)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import pyoracle as orc
            from concurrent.futures import ThreadPoolExecutor
            rng = np.random.default_rng(1)
            pick = rng.choice(len(self.i1), size=min(cpu_sample, len(self.i1)), replace=False)
            nthreads = rank_threads(self.world)
            f = lambda k: orc.gotoh_score_prof(self.profs[self.i1[k]], self.profs[self.i2[k]], 1, 1, SCORE)  # noqa: E731
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=nthreads) as ex:
                want = list(ex.map(f, pick))
            cdt = time.perf_counter() - t0
            t1 = time.perf_counter()
            f(pick[0])
            c1 = time.perf_counter() - t1
            got = self.scores[:len(self.i1)].cpu().numpy()  # this rank's slice of the matrix (the sample indexes it)
            line["cpu_baseline"] = {"value": round(len(pick) * mf * mf / cdt / 1e9, 4), "unit": "GCUPS", "cores": min(nthreads, len(pick)), "kind": "port",
                                    "sample": "%d of the same pairs through the oracle's gotohScore (profile x profile), one pair per thread, %.1f s" % (len(pick), cdt),
                                    "single_thread": {"value": round(mf * mf / c1 / 1e9, 4), "unit": "GCUPS"}}
            line["parity_checked"] = {"pairs": int(len(pick)), "bit_identical": bool(all(int(got[k]) == w for k, w in zip(pick, want)))}
        return line


# =====================================================================================================================
class SeedExtendLeg:
    """configs[3]: `total` 1 kb traces (BASELINE: one million) sampled from a synthetic genome (GRCh38 chr22 -- 50.8 Mb -- is not
    available offline; a random genome of `genome_mb` Mb stands in), host k-mer seeding (getReferenceSlice, fmindex.h:236-326, on this
    rank's share of the host threads) + device extend (tracyhip_align_traces with job.oriented: one 16-bit score sweep, the
    preliminary and the final alignment on their bands, trimReferenceSlice).
    Traces are sharded over the ranks by contiguous blocks.  The k-mer table is built ONCE: rank 0 writes the index file
    (`tracy index`, seed.hpp GenomeIndex::save), every rank maps it read-only -- one copy in the node's page cache."""

    def __init__(self, total, genome_mb, trace_len, rank, world, dev, dist=None):
        import tempfile
        import tracy_amd
        from tracy_amd import capi, hostlib
        from tracy_amd.shard import shard_range
        self.capi, self.total, self.mf, self.rank, self.world, self.dev = capi, total, trace_len, rank, world, dev
        self.threads = rank_threads(world)
        rng = np.random.default_rng(22)  # the same genome and traces on every rank; a rank keeps its block
        n = self.gn = int(genome_mb * 1e6)
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        seq = lut[rng.integers(0, 4, size=n, dtype=np.uint8)]
        shared = os.path.join(tempfile.gettempdir(), "tracy_bench_index_%s_%d" % (os.environ.get("MASTER_PORT", "solo"), os.getppid() if world > 1 else os.getpid()))
        self.tmp = shared
        ipath = os.path.join(shared, "genome.tidx")
        t0 = time.perf_counter()
        if rank == 0:
            os.makedirs(shared, exist_ok=True)
            gpath = os.path.join(shared, "genome.fa")
            with open(gpath, "wb") as f:
                f.write(b">chrSyn\n")
                f.write(seq.tobytes())
                f.write(b"\n")
            built = hostlib.Genome(gpath, 15, self.threads)
            built.save(ipath + ".tmp")
            os.replace(ipath + ".tmp", ipath)
            built.close()
            os.remove(gpath)
        if dist is not None:
            dist.barrier()
        self.genome = hostlib.Genome(ipath, 15, self.threads)  # mapped
        self.index_s = time.perf_counter() - t0
        self.index_bytes = os.path.getsize(ipath)
        lo, hi = shard_range(total, rank, world)
        mf = trace_len
        starts_all = rng.integers(0, n - mf - 50, size=total)
        errs = np.random.default_rng(23 + rank)
        self.starts = starts_all[lo:hi]
        nt = self.nt = hi - lo
        # traces: every other one reads the reverse strand; 1 % substitutions; peaked profile columns.  Built block by block (the blocks the
        # step hands to the device one after the other): a million traces are 24 GB of profiles, and no temporary of that size is wanted
        comp = np.array([3, 2, 1, 0], dtype=np.uint8)
        lut_inv = np.zeros(256, np.uint8)
        lut_inv[lut] = np.arange(4, dtype=np.uint8)
        self.lo = lo
        self.CHUNKS = CH = max(4, (nt + 62499) // 62500)  # blocks of at most 62 500 traces (31 250 at the per-GPU size of an 8-GPU job)
        self.block = [(nt * b // CH, nt * (b + 1) // CH) for b in range(CH)]
        self.profs_b, self.packed = [], []
        win_idx = np.arange(mf, dtype=np.int64)
        for blo, bhi in self.block:
            kb = bhi - blo
            idx = self.starts[blo:bhi, None].astype(np.int64) + win_idx[None, :]
            c = lut_inv[seq[idx]]                                   # [kb][mf] codes of the forward strand
            odd = ((lo + blo + np.arange(kb)) % 2).astype(bool)
            c[odd] = comp[c[odd][:, ::-1]]
            flip = errs.random((kb, mf)) < 0.01
            c = np.where(flip, (c + 1) % 4, c).astype(np.uint8)
            pr = np.zeros((kb, 6, mf), np.float32)
            pr[:, :4, :] = 0.02
            for code in range(4):
                pr[:, code, :][c == code] = 0.94
            self.profs_b.append(pr)
            cons = lut[c]
            self.packed.append(hostlib.Genome.pack_consensus([cons[k].tobytes() for k in range(kb)]))
        self.ctx = tracy_amd.Context(dev.index or 0)
        self.lib = capi.lib()
        self.seed_out = [None] * self.CHUNKS

    def run(self, dist, steps, warmup, cpu_sample=64):
        capi, ctx = self.capi, self.ctx
        seed_s = ext_s = 0.0
        res = sd = ok = None
        self.lib.tracyhip_timing_enable(ctx._h, 0)
        for it in range(warmup + steps):
            if it == warmup:
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                self.lib.tracyhip_timing_enable(ctx._h, 1)
                self.lib.tracyhip_timing_reset(ctx._h)
                t_all = time.perf_counter()
            # The batch goes through in CHUNKS blocks: while the device extends block b (tracyhip_align_traces_async, queued on the
            # context), the host seeds block b + 1 -- the host stage of configs[3] hides behind the device stage and vice versa.
            t0 = time.perf_counter()
            CH = self.CHUNKS
            preps, sds, oks = [], [], []
            step_seed = 0.0
            for b in range(CH):
                blo, bhi = self.block[b]
                ts0 = time.perf_counter()
                # (the block as a C caller holds it, packed once; result buffers of the same block of the previous step reused --
                # by then its extend has finished and its results have been read)
                sdb = self.genome.seed_packed(self.packed[b], 50, 50, 3, 1000, self.threads, out=self.seed_out[b])
                self.seed_out[b] = sdb
                step_seed += time.perf_counter() - ts0
                okb = np.nonzero(sdb["status"] == 1)[0]
                # packed host buffers as a C caller holds them: the profile block and the padded window block are handed over in
                # place, offsets / lengths select the anchored traces
                pp = capi.PackedSeqs([], capi.SEQ_PROFILE)
                pp.count, pp.data = len(okb), self.profs_b[b]
                pp.offset = (okb.astype(np.uint64) * np.uint64(6 * self.mf))
                pp.length = np.full(max(len(okb), 1), self.mf, np.uint32)
                cap = sdb["slices_2d"].shape[1]
                pw = capi.PackedSeqs([], capi.SEQ_CHAR)
                pw.count, pw.data = len(okb), sdb["slices_2d"]
                pw.offset = (okb.astype(np.uint64) * np.uint64(cap))
                pw.length = np.ascontiguousarray(sdb["slice_len"][okb], dtype=np.uint32)
                prep = capi.PreparedAlign(pp, pw, SCORE, 50, 50, oriented=np.ascontiguousarray(sdb["forward"][okb], dtype=np.uint8))
                ctx.align_traces_async(prep.job, prep.prm, prep.out, capi.MEM_HOST)
                preps.append(prep); sds.append(sdb); oks.append(okb + blo)
            ctx.synchronize()
            t3 = time.perf_counter()
            t1, t2 = t0 + step_seed, t0 + step_seed  # (seeding time of the step; the extend overlaps it)
            # results of the step, block after block
            rs = [p_.results() for p_ in preps]
            res = {k: (np.concatenate([r[k] for r in rs]) if k != "btr" else sum((r["btr"] for r in rs), [])) for k in ("score_final", "slice_len", "ref_pos", "btr")}
            ok = np.concatenate(oks)
            sd = {"pos": np.concatenate([sds[b]["pos"] for b in range(CH)]), "slice_len": np.concatenate([sds[b]["slice_len"] for b in range(CH)]),
                  "forward": np.concatenate([sds[b]["forward"] for b in range(CH)]), "status": np.concatenate([sds[b]["status"] for b in range(CH)])}
            self._win = lambda i: next(sds[b]["slices_2d"][i - self.block[b][0], :int(sds[b]["slice_len"][i - self.block[b][0]])].tobytes()
                                       for b in range(CH) if self.block[b][0] <= i < self.block[b][1])
            self._prof = lambda i: next(self.profs_b[b][i - self.block[b][0]] for b in range(CH) if self.block[b][0] <= i < self.block[b][1])
            if it >= warmup:
                seed_s += step_seed
                ext_s += (t3 - t0) - step_seed  # what the extend adds on top of the seeding it overlaps with
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t_all
        self.lib.tracyhip_timing_enable(ctx._h, 0)
        timers = read_timers(self.lib, ctx)
        self.genome.close()  # the mapped index file goes away with the leg
        if dist is not None:
            dist.barrier()
        if self.rank == 0:
            import shutil
            shutil.rmtree(self.tmp, ignore_errors=True)
        mf = self.mf
        win_len = sd["slice_len"][ok].astype(np.int64)
        cells = int(((mf - 100) * win_len).sum() * 2 + (mf * res["slice_len"].astype(np.int64)).sum())
        truth = self.starts[ok] - np.where((self.lo + ok) % 2 == 0, 50, 0)
        placed = int((np.abs(sd["pos"][ok].astype(np.int64) + res["ref_pos"].astype(np.int64) - truth) <= 60).sum())
        dt, seed_s, ext_s = max_over_ranks(dist, self.dev, [dt, seed_s, ext_s])
        cells_all, ok_all, placed_all, nt_all = sum_over_ranks(dist, self.dev, [float(cells), float(len(ok)), float(placed), float(self.nt)])
        if self.rank != 0:
            return None
        roof = kernel_block("score", "gotoh_ckpt_kernel<K,QP,narrow> (one orientation: the window arrives oriented by the seeds)", timers["score"], steps,
                            ops_per_cell=8.0, traffic_key=[r"gotoh_ckpt_kernel<"], traffic_glob="r[0-9][0-9]_pmc_hbm_seedextend.json")
        roof["ms_per_step"] = {k: round(timers[k]["ms"] / steps, 3) for k, _ in TIMERS if timers[k]["ms"] > 0}
        roof["note"] = "four blocks per step, each extended asynchronously while the host seeds the next one"
        from bench import usable_cores
        line = {"metric": "traces/s (host k-mer seeding + device Gotoh extend, end to end)", "value": round(ok_all * steps / dt, 1), "unit": "traces/s",
                "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps, "warmup": warmup, "n_gpus": self.world, "scaling": "strong",
                "seed_traces_per_s": round(nt_all * steps / seed_s, 1),
                # seeding is host work (fmindex.h:203-326): what one host thread seeds, what one GPU extends per second of its kernels'
                # time, and how many GPUs the threads of THIS run (one rank's share of the node) could keep busy
                "seed_traces_per_s_per_thread": round(nt_all * steps / seed_s / max(self.threads * self.world, 1), 1),
                "extend_traces_per_s_per_gpu": round(ok_all / self.world * steps / max(sum(timers[k]["ms"] for k, _ in TIMERS) * 1e-3, 1e-9), 1),
                "n_gpus_fed_at_this_host": round((nt_all * steps / seed_s) / max(ok_all / self.world * steps / max(sum(timers[k]["ms"] for k, _ in TIMERS) * 1e-3, 1e-9), 1e-9), 2),
                "host_threads_needed_to_feed_one_gpu": round(max(ok_all / self.world * steps / max(sum(timers[k]["ms"] for k, _ in TIMERS) * 1e-3, 1e-9), 1e-9) / max(nt_all * steps / seed_s / max(self.threads * self.world, 1), 1e-9), 1),
                "extend_ms_not_hidden_per_step": round(ext_s / steps * 1e3, 2), "extend_kernel_gcups": round(cells_all * steps / max(sum(timers[k]["ms"] for k in ("score", "trace", "band", "walk")) * 1e-3, 1e-9) / 1e9, 1), "host_threads_per_rank": self.threads, "index_build_and_map_s": round(self.index_s, 2), "index_file_mb": round(self.index_bytes / 1e6, 1),
                "index": "built once (rank 0), written with GenomeIndex::save, mapped read-only by every rank",
                "anchored": int(ok_all), "traces": int(nt_all), "placed_within_60bp_of_truth": int(placed_all),
                "dtype": "int16 (score sweep) / int32 (tracebacks); seeding: 2-bit k-mers on the host",
                "config": {"workload": "configs[3]: %d traces of %d bases (%d per rank, in blocks of <= 62 500) vs a %.0f Mb synthetic genome (chr22-sized), k = 15, "
                                       "window = trace + 2 x 1000, traces sharded over %d rank(s), one mapped k-mer table per node"
                                       % (int(nt_all), mf, int(nt_all) // self.world, self.gn / 1e6, self.world),
                           "traces_total": int(nt_all), "blocks_per_rank": self.CHUNKS},
                "data": "synthetic (GRCh38 chr22 is not available offline); host-staged buffers: upload of profiles / windows and download of results included",
                "roofline": roof}
        if cpu_sample > 0:  # (rank 0; at N > 1 on its share of the host cores)
            for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
                if p not in sys.path:
                    sys.path.insert(0, p)
            import pyoracle as orc
            from concurrent.futures import ThreadPoolExecutor
            # the sample: spread over the whole batch (every block of the step takes part)
            npick = min(cpu_sample, len(ok))
            pick = [int(x) for x in np.unique(np.linspace(0, len(ok) - 1, npick).astype(np.int64))] if npick else []

            def one(k):  # sage.h:217-221, 258-260, 311 with a window that arrives oriented: two Gotoh calls and the trim
                i = ok[k]
                prof = self._prof(i)
                win = self._win(i)
                trimmed = np.ascontiguousarray(prof[:, 50:mf - 50])
                pref = orc.create_profile_str(win)
                sc1, btr1 = orc.gotoh_prof(trimmed, pref, 1, 0, SCORE)
                r0, r1 = orc.create_alignment_prof(btr1, trimmed, pref)
                ri, risize, pos_add, _ = orc.trim_reference_slice(r0, r1, 50, 50, len(win), bool(sd["forward"][i]))
                sc2, btr2 = orc.gotoh_prof(prof, orc.create_profile_str(win[ri:ri + risize]), 1, 0, SCORE)
                return sc2, btr2, pos_add
            nthreads = self.threads
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=nthreads) as ex:
                want = list(ex.map(one, pick))
            cdt = time.perf_counter() - t0
            okp = all(int(res["score_final"][k]) == w[0] and res["btr"][k] == w[1] and int(res["ref_pos"][k]) == w[2] for k, w in zip(pick, want))
            line["cpu_baseline"] = {"value": round(len(pick) / cdt, 2), "unit": "traces/s (extend only; the seeding is host work in both)", "cores": min(nthreads, len(pick)),
                                    "kind": "port", "sample": "%d of the same anchored traces through the oracle's two Gotoh calls + trimReferenceSlice, %.1f s" % (len(pick), cdt)}
            line["parity_checked"] = {"traces": len(pick), "bit_identical": bool(okp)}
        return line


# =====================================================================================================================
class CliLeg:
    """The product end to end (north star: "the `tracy align` / `decompose` CLI and JSON output stay unchanged ... ABIF parsing,
    basecalling ... stay on the host"): synthetic ABIF traces + one FASTA window per trace on disk, `tracy_amd_cli align --batch`
    and `decompose --batch` (sage.h:58-356, indigo.h:42-455: readab abif.h:286-405, basecall :408-511, createProfile, the device
    pipelines, the .json / .txt / .fa writers of json.h:197-381), wall time of the whole command with the CLI's own split of it
    (TRACY_AMD_CLI_TIMERS).  The files are written by the build's ABIF writer before the clock starts."""

    def __init__(self, ntraces, rank, world, dev, workdir=None):
        import tempfile
        self.nt, self.rank, self.world, self.dev = ntraces, rank, world, dev
        # where the files live: a tmpfs when the box has one with room for a run's ~1 MB of text per trace (a fresh file on a block
        # device costs its block allocation, ~1 ms per 400 KB file on this image's disk -- the file system's time, not the command's);
        # --cli-workdir puts them anywhere else
        if workdir is None:
            import shutil
            shm = "/dev/shm"
            need = 3 * (1 << 20) * ntraces
            base = shm if os.path.isdir(shm) and os.access(shm, os.W_OK) and shutil.disk_usage(shm).free > need else None
            self.tmp = tempfile.mkdtemp(prefix="tracy_cli_bench_", dir=base)
        else:
            os.makedirs(workdir, exist_ok=True)
            self.tmp = tempfile.mkdtemp(prefix="tracy_cli_bench_", dir=workdir)
        self.fs = self.filesystem_of(self.tmp)
        self.cli = os.path.join(ROOT, "tracy_amd", "bin", "tracy_amd_cli")

    @staticmethod
    def filesystem_of(path):
        best, kind = "", "unknown"
        try:
            for ln in open("/proc/mounts"):
                f = ln.split()
                if len(f) >= 3 and path.startswith(f[1]) and len(f[1]) > len(best):
                    best, kind = f[1], f[2]
        except OSError:
            pass
        return kind

    def make_files(self, cmd, n, mf):
        from tracy_amd import hostlib
        d = os.path.join(self.tmp, cmd)
        os.makedirs(d, exist_ok=True)
        t0 = time.perf_counter()
        b = hostlib.synth_decompose_batch(7000 if cmd == "align" else 8000, self.nt, n, mf, 0, mix=1)
        rows = []
        q40 = np.full(mf, 40, np.uint8)
        for i in range(self.nt):
            tp = os.path.join(d, "t%06d.ab1" % i)
            rp = os.path.join(d, "r%06d.fa" % i)
            hostlib.write_abif(tp, np.minimum(b["signal"][i], 32000), b["bcpos"][i], b"N" * mf, q40)
            with open(rp, "wb") as f:
                f.write(b">win%06d\n" % i)
                f.write(b["refs"][i].tobytes())
                f.write(b"\n")
            rows.append("%s\t%s\t%s\n" % (tp, rp, os.path.join(d, "o%06d" % i)))
        man = os.path.join(d, "manifest.tsv")
        open(man, "w").write("".join(rows))
        return man, d, time.perf_counter() - t0, b

    def run_cli(self, cmd, man):
        import subprocess
        env = dict(os.environ, TRACY_AMD_CLI_TIMERS="1")
        t0 = time.perf_counter()
        p = subprocess.run([self.cli, cmd, "--batch", man, "-d", str(self.dev.index or 0)], capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t0
        split = {}
        for ln in p.stderr.splitlines():
            if ln.startswith("timers:"):
                tok = ln.split()[1:]
                split = {tok[i]: float(tok[i + 1]) for i in range(0, len(tok) - 1, 2)}
        if p.returncode not in (0, 2):
            raise RuntimeError("tracy_amd_cli %s --batch failed (%d): %s" % (cmd, p.returncode, p.stderr[-400:]))
        return dt, split, p.returncode

    def cpu_baseline(self, cmd, b, sample):
        """the oracle's chain on the same traces from the file on: the reference's own abif.h reader + basecall (oracle/_ref), the
        oracle's createProfile and its sage.h / indigo.h chain, one trace per thread (writers not included)"""
        for p_ in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            if p_ not in sys.path:
                sys.path.insert(0, p_)
        from concurrent.futures import ThreadPoolExecutor
        import pyoracle as orc
        from bench import usable_cores
        from indigo_oracle import decompose_trace
        from sage_oracle import align_trace
        nthreads = usable_cores()
        ns_ = min(sample, self.nt)

        def one(i):
            sig, pos = np.minimum(b["signal"][i], 32000), b["bcpos"][i]
            pri, sec = orc.basecall(sig, pos, 0.33)[:2]
            if cmd == "align":
                prof = orc.create_profile_trace(sig, pos, pri, sec, 0, 0)
                return align_trace(prof, b["refs"][i].tobytes(), SCORE)["score_final"]
            return decompose_trace(sig, pos, pri, sec, b["refs"][i].tobytes(), SCORE)["status"]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nthreads) as ex:
            list(ex.map(one, range(ns_)))
        dt = time.perf_counter() - t0
        return {"value": round(ns_ / dt, 2), "unit": "traces/s", "cores": min(nthreads, ns_), "kind": "port",
                "sample": "%d of the same traces through the oracle's basecall + createProfile + %s chain (one trace per thread; file parsing and "
                          "writers not included), %.1f s" % (ns_, "sage.h" if cmd == "align" else "indigo.h", dt)}

    def run(self, dist, cpu_sample=0):
        import shutil
        from bench import usable_cores
        out = {"metric": "traces/s, `tracy_amd_cli --batch` end to end (ABIF files in, JSON / txt / fa files out)", "unit": "traces/s",
               "n_gpus": 1, "host_threads": usable_cores(), "data": "synthetic ABIF files written by the build's own writer (not timed)",
               "files_on": "%s (%s)" % (self.fs, os.path.dirname(self.tmp)),
               "config": {"workload": "%d traces of 1000 bases per command: `align` vs a 10 kb FASTA window each (configs[1] through the CLI), "
                                      "`decompose` vs a 3 kb window each (configs[2] through the CLI)" % self.nt}}
        try:
            for cmd, n in (("align", 10000), ("decompose", 3000)):
                man, d, prep_s, b = self.make_files(cmd, n, 1000)
                dt, split, rc = self.run_cli(cmd, man)
                written = sum(1 for f in os.listdir(d) if f.endswith(".json"))
                rec = {"traces_per_s": round(self.nt / dt, 1), "wall_s": round(dt, 3), "split_s": split, "json_files_written": written,
                       "exit_code": rc, "files_prepared_s": round(prep_s, 1)}
                if split:
                    cpu = {k[4:]: round(v, 3) for k, v in split.items() if k.startswith("cpu_")}
                    split = {k: v for k, v in split.items() if not k.startswith("cpu_")}
                    rec["split_s"] = split
                    # seconds inside the stages by what they were spent on: host-thread seconds summed over the threads for the prep /
                    # writer work, wall seconds of the calling thread for pack / device_call / unpack / variants
                    rec["thread_seconds_by_phase"] = cpu
                    stages = {k: v for k, v in split.items() if k.endswith("_s") and k != "wall_s"}
                    host = split.get("read_basecall_profile_s", 0.0) + split.get("writers_s", 0.0)
                    rec["host_stage_seconds_over_wall"] = round(host / max(dt, 1e-9), 3)
                    rec["first_bottleneck"] = max(stages, key=lambda k: stages[k])
                    rec["stages_overlap"] = round(sum(stages.values()) / max(split.get("wall_s", dt), 1e-9), 2)
                    rec["peak_rss_mb"] = split.get("peak_rss_mb")
                    # the command's wall time that is not its stages: loading the HIP runtime, the manifest, tearing the process down
                    rec["outside_stages_s"] = round(dt - split.get("wall_s", dt), 3)
                    rec["note"] = ("the manifest runs in blocks of 2000 traces through three stages in flight: host threads read / basecall / profile block "
                                   "k + 1 and write the files of block k - 1 while the device works on block k; split_s = the seconds each stage was busy "
                                   "(they overlap: stages_overlap = their sum / wall), peak RSS is bounded by the blocks in flight; device_s = packing a "
                                   "block, tracyhip_*_traces with host buffers (PCIe both ways), unpacking, alignment rows; gpu_init_s = creating the "
                                   "context (HIP start-up), once per command")
                if cpu_sample > 0:
                    rec["cpu_baseline"] = self.cpu_baseline(cmd, b, cpu_sample)
                out[cmd] = rec
                del b
                shutil.rmtree(d, ignore_errors=True)
            out["value"] = out["align"]["traces_per_s"]
        finally:
            shutil.rmtree(self.tmp, ignore_errors=True)
        return out
