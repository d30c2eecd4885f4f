#!/usr/bin/env python
"""bench.py -- `tracy align` hot path (sage.h:191-311) on N MI355X, one process per GPU.

A step = one pass of tracyhip_align_traces over this rank's batch of synthetic traces (BASELINE.json
configs[1]: 10k x 1 kb traces vs 10 kb reference windows per GPU; inputs resident in HBM before the
timed region).  Per trace: 2 score-only Gotoh DPs (forward / reverse-complement reference), 1 traceback
DP of the trimmed profile (taken by its two ends: an origin-tracking sweep over the sub-window the score sweep certifies),
trimReferenceSlice, 1 traceback DP of the full profile vs the trimmed slice.
Weak scaling: the per-GPU batch is fixed; traces shard by index with no data-path collective; every timed step ends with the job's
final gather to rank 0 in its two halves (tracy_amd/shard.py ResultGather): the fixed-size record of every trace in one
collective, then the traceback strings -- packed on the device (tracyhip_pack_ragged_multi) -- in one grouped exchange sized from the
records' length column.  `gather_ms_per_step` is that part of `ms_per_step`.

Legs (each: W warm-up + K timed steps between barriers): (1) the headline -- one lane, exact gsFwd AND gsRev (`value`, `roofline`: what
a zero-initialised job gets): the strand a k-mer vote does not pick is swept in full, the voted one by its first 128 rows over the
window and a certified band below them (front.h; swept in full where the certificate fails); (2) the strand-by-certificate mode
(only the winner's score exact -- all that tracy's output carries; what tracy_amd_cli runs): 128-row prefixes of both strands, the
voted one continued on its band, the other one decided by its bound; (3), (4) the same two on `--lanes-leg` chunks of the batch in
flight.  Legs 2-4 are checked to return the headline leg's alignments and are reported beside it;
`--certificate-leg 0 --lanes-leg 0` runs the headline alone (what the rocprofv3 passes under profiles/ use).

Beside the headline the default run also times BASELINE.json configs[2] (`tracy decompose`, 100 000 traces sharded over the
ranks) configs[4] (all-pairs profile x profile scoring of 1000 traces, pair list sharded over the ranks, score slices
all-gathered) and configs[3] in miniature (host seeding + device extend) -- tools/legs.py -- each with its own roofline, cpu_baseline and in-run parity sample, reported as the
`decompose`, `allpairs` and `seedextend` objects of the same line (`--workload align|decompose|allpairs|seedextend` runs one of them alone, which
is what the rocprofv3 passes under profiles/ use).  `value` is always the align headline.

`--gpus N` without a torch.distributed environment starts the N ranks itself (torch.distributed.run, 127.0.0.1) after checking
that the box has N GPUs; every rank asserts that the process group really has N members.

Prints ONE JSON line on rank 0, cut to what the driver's record keeps (finish_line: < 8 KB, the figures of every leg as scalars at the
top level and in `config` / `roofline`); the unabridged line goes to --full-line (default gpurun_out/bench_line_full.json).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCORE = (3, -5, -10, -4)  # match, mismatch, gapopen, gapext (sage.h:79-82)
TRIM = 50                 # trimLeft = trimRight = 50 (sage.h:88-89)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured copy ceiling)


def cpu_baseline(profs, refs, nthreads, budget_traces):
    """The oracle (CPU restatement of the reference path, `kind: port`) timed on this host: the sage.h chain in C, one
    trace per C thread (oracle/tracy_oracle_chain.c).  Only this leg may touch oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    pyoracle.lib()
    n = min(budget_traces, len(profs))
    p = np.ascontiguousarray(profs[:n], dtype=np.float32)
    r = np.ascontiguousarray(refs[:n], dtype=np.uint8)
    t0 = time.perf_counter()
    res, cells = pyoracle.sage_chain_batch(p, r, SCORE, TRIM, TRIM, nthreads)
    dt = time.perf_counter() - t0
    return cells / dt / 1e9, n, dt, res


def usable_cores():
    """cores this process can really run on: the affinity mask, capped by the cgroup CPU quota of the container"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """--gpus N, no torch.distributed environment: become N ranks of one node (one process per GPU, RCCL over xGMI)"""
    if not args.stub and not args.share_device:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but this box exposes %d GPU(s); refusing to report a %d-GPU number from fewer devices"
                             % (args.gpus, have, args.gpus))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def stub_rank(args, rank, world):
    """Launcher self-test (tests/test_bench_launcher.py): the rank / process-group / barrier / max-over-ranks / one-line path of
    this file on the gloo backend with a step that does no device work.  NOT a measurement: the line says "stub": true."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if dist.get_world_size() != args.gpus:
        raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    x = torch.zeros(4)

    def step():
        x.add_(1.0)
    for _ in range(args.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dist.barrier()
    tm = torch.tensor([time.perf_counter() - t0, float(rank + 1)], dtype=torch.float64)
    tmax, tsum = tm.clone(), tm.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"metric": "launcher-selftest", "stub": True, "value": 0.0, "unit": "none", "n_gpus": world, "rccl_ranks": 0,
                          "backend": "gloo", "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(tmax[0]) / max(args.steps, 1) * 1e3, 4),
                          "rank_sum": float(tsum[1])}))
    dist.destroy_process_group()


DRIVER_LINE_LIMIT = 7800  # bytes: the driver's record keeps the parsed contract keys, `config`, the scalars of `roofline`, `cpu_baseline` and an 8 KB tail


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def finish_line(line, args):
    """The line the driver keeps: every number a reader needs as a scalar where the driver's record retains it
    (top level, `config`, `roofline`), the legs' objects cut down to their figures, the prose moved to DESIGN.md -- under 8 KB.  The
    unabridged line (every leg's detail, as printed until round 5) goes to --full-line / gpurun_out/bench_line_full.json."""
    full = json.loads(json.dumps(line))
    path = args.full_line
    if path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        path = os.path.join(ROOT, "gpurun_out", "bench_line_full.json")
    if path:
        try:
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
        except OSError:
            pass
    out = dict(line)
    scal = {}
    d = line.get("decompose")
    if isinstance(d, dict) and "error" not in d:
        sb = d.get("small_batch") or {}
        scal.update({"decompose_ms_per_step": d.get("ms_per_step"), "decompose_traces_per_s": d.get("value"),
                     "small_batch_ms": sb.get("ms_per_step"), "small_batch_ratio": sb.get("vs_eighth_of_the_full_step"),
                     "decompose_gathered_bytes_per_step": d.get("gathered_bytes_per_step")})
        r = d.get("roofline") or {}
        out["decompose"] = dict(_pick(d, ("value", "unit", "ms_per_step", "gcups", "steps", "n_gpus", "scaling", "traces_ok", "gathered_bytes_per_step", "gather_ms_per_step", "gather_checked", "parity_checked")),
                                small_batch=_pick(sb, ("traces", "ms_per_step", "two_lanes_ms_per_step", "vs_eighth_of_the_full_step")),
                                strand_by_certificate_ms=(d.get("strand_by_certificate") or {}).get("ms_per_step"),
                                two_lanes_ms=(d.get("lanes") or {}).get("ms_per_step"),
                                roofline=dict(_pick(r, ("bound", "timer", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "kernel_gcups", "share_of_kernel_time", "ms_per_step")),
                                              other={k: _pick(v, ("achieved", "frac", "traffic", "avg_launch_ms", "kernel_gcups")) for k, v in (r.get("other_kernels") or {}).items()}),
                                pipeline=_pick(d.get("pipeline") or {}, ("stream_ordered", "host_syncs_per_call", "traces_to_host_planned_tiers", "traces_per_rank")),
                                cpu_baseline=_pick(d.get("cpu_baseline") or {}, ("value", "unit", "cores", "kind")))
    a = line.get("allpairs")
    if isinstance(a, dict) and "error" not in a:
        scal["allpairs_gcups"] = a.get("value")
        out["allpairs"] = dict(_pick(a, ("value", "unit", "pairs", "ms_per_step", "n_gpus", "scaling", "parity_checked")),
                               roofline=_pick(a.get("roofline") or {}, ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "kernel_gcups")),
                               cpu_baseline=_pick(a.get("cpu_baseline") or {}, ("value", "unit", "cores", "kind")))
    e = line.get("seedextend")
    if isinstance(e, dict) and "error" not in e:
        scal.update({"seedextend_traces_per_s": e.get("value"), "seed_traces_per_s_per_thread": e.get("seed_traces_per_s_per_thread")})
        out["seedextend"] = dict(_pick(e, ("value", "unit", "ms_per_step", "n_gpus", "scaling", "seed_traces_per_s", "seed_traces_per_s_per_thread", "extend_traces_per_s_per_gpu",
                                           "n_gpus_fed_at_this_host", "host_threads_needed_to_feed_one_gpu", "extend_ms_not_hidden_per_step", "host_threads_per_rank", "anchored", "traces",
                                           "placed_within_60bp_of_truth", "parity_checked")),
                                 roofline=_pick(e.get("roofline") or {}, ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "kernel_gcups")),
                                 cpu_baseline=_pick(e.get("cpu_baseline") or {}, ("value", "unit", "cores", "kind")))
    c = line.get("cli")
    if isinstance(c, dict) and "error" not in c:
        sub = {}
        for cmd in ("align", "decompose"):
            x = c.get(cmd) or {}
            scal["cli_%s_traces_per_s" % cmd] = x.get("traces_per_s")
            sub[cmd] = dict(_pick(x, ("traces_per_s", "wall_s", "json_files_written", "exit_code", "first_bottleneck", "peak_rss_mb")),
                            cpu_baseline_traces_per_s=(x.get("cpu_baseline") or {}).get("value"))
        out["cli"] = dict(_pick(c, ("value", "unit", "host_threads")), **sub)
    scal = {k: v for k, v in scal.items() if v is not None}
    out.update(scal)
    out["config"] = dict(line["config"], **scal)
    out["full_line"] = os.path.relpath(path, ROOT) if path else None
    for drop in ("cli", "seedextend", "allpairs", "lanes", "strand_by_certificate", "pipeline"):  # (never needed with the fields above; a guard, not a plan)
        if len(json.dumps(out)) <= DRIVER_LINE_LIMIT:
            break
        out.pop(drop, None)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="all", choices=("all", "align", "decompose", "allpairs", "seedextend", "cli"),
                    help="all: the align headline + the decompose and all-pairs legs; one name: that workload alone (profiling)")
    ap.add_argument("--decompose-traces", type=int, default=100000, help="configs[2]: traces of the whole decompose job (sharded over the ranks)")
    ap.add_argument("--decompose-steps", type=int, default=3)
    ap.add_argument("--allpairs-traces", type=int, default=1000, help="configs[4]: traces of the all-pairs job (pair list sharded over the ranks)")
    ap.add_argument("--allpairs-steps", type=int, default=3)
    ap.add_argument("--seedextend-traces", type=int, default=1000000,
                    help="configs[3]: traces of the seed + extend job IN ALL (BASELINE: 1M traces), sharded over the ranks -- one GPU takes the million")
    ap.add_argument("--no-process-group", action="store_true", help="a rank started by hand: no torch.distributed process group of one (default: nccl, world size 1)")
    ap.add_argument("--seedextend-genome-mb", type=float, default=50.0, help="configs[3]: size of the synthetic genome (GRCh38 chr22 is 50.8 Mb)")
    ap.add_argument("--seedextend-steps", type=int, default=1)
    ap.add_argument("--cli-workdir", default=None, help="the CLI leg: directory for its input and output files (default: /dev/shm when it has room, else the system's temporary directory)")
    ap.add_argument("--cli-traces", type=int, default=10000, help="the CLI leg: ABIF files per command (`align --batch`, `decompose --batch`); 0 = skip")
    ap.add_argument("--extra-legs", type=int, default=1, help="decompose: also time the strand-certificate and two-lane legs")
    ap.add_argument("--stub", action="store_true", help="launcher self-test on gloo with a step that does no device work (not a measurement)")
    ap.add_argument("--share-device", action="store_true",
                    help="TEST MODE (tests/test_gpu_bench_ranks.py): the N ranks of --gpus N all run on GPU 0 and gather over gloo -- the code path of "
                         "a multi-GPU run on a one-GPU box; the line says \"shared_device\": true and is not a measurement")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--traces", type=int, default=10000, help="traces per GPU per step")
    ap.add_argument("--ref-len", type=int, default=10000)
    ap.add_argument("--trace-len", type=int, default=1000)
    ap.add_argument("--lanes", type=int, default=1,
                    help="headline leg: chunks of the batch in flight inside one tracyhip_align_traces call (tracyhip_set_lanes)")
    ap.add_argument("--lanes-leg", type=int, default=2,
                    help="also time the batch split over this many lanes (reported beside the headline; 0/1 = skip)")
    ap.add_argument("--certificate-leg", type=int, default=1,
                    help="1: also time the library's default strand-by-certificate mode after the headline leg (reported beside it)")
    ap.add_argument("--alone-steps", type=int, default=3, help="untimed extra steps with option sweeps_alone: the dominant kernel on a device of its own (0 = skip)")
    ap.add_argument("--full-line", default=None, help="also write the unabridged line (every leg's detail) to this file (default: gpurun_out/bench_line_full.json when that directory exists)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="traces for the CPU baseline (-1: 2 per thread, capped)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or let --gpus N start them)" % (args.gpus, world))
    if args.stub:
        return stub_rank(args, rank, world)
    if args.share_device:
        local = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but %d visible GPU(s)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    if world > 1:  # the library's host workers (descriptor loops between launches): this rank's share of the node's cores, not all of them
        from tools.legs import rank_threads
        os.environ.setdefault("TRACYHIP_HOST_THREADS", str(rank_threads(world)))
    dist = None
    backend = "none"  # no process group: one rank started by hand
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        backend = dist.get_backend()
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    elif not args.no_process_group:
        # one rank started by hand (the driver's N = 1 run): a process group of one all the same, so that the gathers of every leg go
        # through RCCL exactly as they do at N > 1 (a failure to set it up is reported in the line, not fatal)
        try:
            import socket
            import torch.distributed as dist
            with socket.socket() as so_:
                so_.bind(("127.0.0.1", 0))
                port = so_.getsockname()[1]
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", local))
            backend = dist.get_backend()
        except Exception as e:  # noqa: BLE001
            dist = None
            backend = "none (%s: %s)" % (type(e).__name__, str(e)[:120])
        if dist is not None and dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    dev = torch.device("cuda", local)

    def run_extra(which):
        """configs[2] / configs[4] legs (tools/legs.py); a failing leg reports its error instead of taking the headline down"""
        from tools.legs import AllPairsLeg, CliLeg, DecomposeLeg, SeedExtendLeg
        try:
            if which == "cli":  # the product end to end; rank 0 only (the CLI shards over GPUs itself with -d)
                if rank != 0 or args.cli_traces <= 0:
                    return None
                return CliLeg(args.cli_traces, rank, world, dev, workdir=args.cli_workdir).run(dist, cpu_sample=320 if args.cpu_sample != 0 else 0)
            if which == "seedextend":
                leg = SeedExtendLeg(args.seedextend_traces, args.seedextend_genome_mb, 1000, rank, world, dev, dist)
                res = leg.run(dist, args.seedextend_steps, 1, cpu_sample=256 if args.cpu_sample != 0 else 0)
            elif which == "decompose":
                leg = DecomposeLeg(args.decompose_traces, 3000, 1000, rank, world, dev)
                res = leg.run(dist, args.decompose_steps, 1, extra_legs=bool(args.extra_legs), cpu_sample=128 if args.cpu_sample != 0 else 0)
            else:
                leg = AllPairsLeg(args.allpairs_traces, 900, rank, world, dev)
                res = leg.run(dist, args.allpairs_steps, 1, cpu_sample=4096 if args.cpu_sample != 0 else 0)
            leg.ctx.close()
            del leg
            torch.cuda.empty_cache()
            return res
        except Exception as e:  # noqa: BLE001
            if args.workload != "all":
                raise
            return {"error": "%s: %s" % (type(e).__name__, e)}

    extra = {}
    if args.workload in ("decompose", "allpairs", "seedextend", "cli"):
        extra[args.workload] = run_extra(args.workload)
        if rank == 0:
            line = extra[args.workload]
            line["backend"] = backend
            line["rccl_ranks"] = world if backend == "nccl" else 0  # ranks RCCL really connected (gloo: bench.py --share-device, the CPU tests)
            print(json.dumps(line))
        if dist is not None:
            dist.destroy_process_group()
        return

    import tracy_amd
    from tracy_amd import capi, hostlib
    from tracy_amd.shard import ResultGather

    nt, n, mf = args.traces, args.ref_len, args.trace_len
    # ---- synthetic inputs (seeded per trace: seed = 1000 + global trace index), resident in HBM ----
    refs, profs, rev = hostlib.synth_align(1000 + rank * nt, nt, n, mf, 0)
    d_refs = torch.from_numpy(refs).cuda()
    d_profs = torch.from_numpy(profs).cuda()
    pp_off = (np.arange(nt, dtype=np.uint64) * np.uint64(6 * mf))
    pp_len = np.full(nt, mf, dtype=np.uint32)
    rr_off = (np.arange(nt, dtype=np.uint64) * np.uint64(n))
    rr_len = np.full(nt, n, dtype=np.uint32)
    ops_cap = mf + n
    ops_off = (np.arange(nt, dtype=np.uint64) * np.uint64(ops_cap))

    job = capi.AlignJob()
    job.ntraces = nt
    job.profiles = capi.SeqSet(capi.SEQ_PROFILE, d_profs.data_ptr(), pp_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                               pp_len.ctypes.data_as(C.POINTER(C.c_uint32)), nt)
    job.refs = capi.SeqSet(capi.SEQ_CHAR, d_refs.data_ptr(), rr_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                           rr_len.ctypes.data_as(C.POINTER(C.c_uint32)), nt)
    job.trim_left = TRIM
    job.trim_right = TRIM
    # headline leg: both orientation scores exact (the strand the k-mer vote does not pick swept in full, the voted one pruned with a
    # certificate: front.h).  The strand-by-certificate mode (identical alignments, only the winner's score exact, fewer cells) is
    # timed as a second leg below and reported beside it -- it is never `value`.
    job.strand_by_certificate = 0
    r_i32 = {k: torch.zeros(nt, dtype=torch.int32, device=dev) for k in
             ("score_fwd", "score_rev", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final", "ops_len")}
    r_fwd = torch.zeros(nt, dtype=torch.uint8, device=dev)
    r_ops = torch.zeros(nt * ops_cap, dtype=torch.uint8, device=dev)
    out = capi.AlignResult()
    for k, v in r_i32.items():
        setattr(out, k, v.data_ptr())
    out.forward = r_fwd.data_ptr()
    out.ops = r_ops.data_ptr()
    out.ops_offset = ops_off.ctypes.data_as(C.POINTER(C.c_uint64))
    prm = capi.Params(SCORE[0], SCORE[1], SCORE[2], SCORE[3], 1, 0)

    capi.lib().tracyhip_tune_host_allocator()  # (process-wide, opt-in: this process is a benchmark, not somebody else's application)
    ctx = tracy_amd.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    # headline leg: one lane, so that every kernel has the device to itself and its launch duration means what the
    # roofline block says.  tracyhip_set_lanes (chunks of the batch in flight on their own streams) is timed as a further leg.
    ctx.set_lanes(max(1, args.lanes))
    lib = capi.lib()
    REC_KEYS = ("score_fwd", "score_rev", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final", "ops_len")
    gatherer = ResultGather(dist, [nt] * world, ctx) if dist is not None else None  # (weak scaling: every rank holds nt traces)
    gathered = {"out": None, "bytes": 0, "seconds": 0.0}

    def run_call():
        rc = lib.tracyhip_align_traces(ctx._h, C.byref(job), C.byref(prm), capi.MEM_DEVICE, C.byref(out))
        if rc != 0:
            raise RuntimeError("tracyhip_align_traces: %s" % lib.tracyhip_last_error().decode())

    def step():
        run_call()  # (synchronous: the results are complete when it returns)
        t_g = time.perf_counter()
        if dist is not None:
            # the final gather, both halves: the fixed-size record of every trace in one collective, then the traceback strings
            # (sage.h:311's alignment) packed on the device and shipped in one exchange sized from the records' ops_len column
            rec = torch.stack([r_i32[k] for k in REC_KEYS] + [r_fwd.to(torch.int32)], dim=1)
            gathered["out"] = gatherer.gather(rec, [(r_ops, ops_cap, REC_KEYS.index("ops_len"))])
            gathered["bytes"] = gatherer.bytes_last
            gathered["seconds"] += time.perf_counter() - t_g  # (the pack ends with a synchronisation: host time = the gather's share of the step on this rank)

    def read_timers():
        kt = capi.KernelTiming()
        res = {}
        for name, which in (("score", 0), ("trace", 1), ("walk", 2), ("band", 3), ("prefix", 4), ("origin", 5), ("front", 9)):
            lib.tracyhip_timing_get(ctx._h, which, C.byref(kt))
            res[name] = dict(ms=kt.ms, launches=int(kt.launches), cells=int(kt.cells), bytes=int(kt.bytes))
        return res

    def timed_leg():
        for _ in range(args.warmup):
            step()
        gathered["seconds"] = 0.0  # (the first gather sets the communicator up)
        lib.tracyhip_timing_enable(ctx._h, 1)
        lib.tracyhip_timing_reset(ctx._h)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        lib.tracyhip_timing_enable(ctx._h, 0)
        return dt, read_timers()

    elapsed, rl = timed_leg()
    gather_ms = gathered["seconds"] / max(args.steps, 1) * 1e3  # (rank 0's share of a timed step spent behind the library call)
    call_stats = ctx.last_call_stats()
    slice_len = r_i32["slice_len"].cpu().numpy().astype(np.int64)
    elapsed_cert, rl_cert = 0.0, None
    if args.certificate_leg:
        exact_final = r_i32["score_final"].clone()
        exact_ops = r_ops.clone()
        job.strand_by_certificate = 1
        elapsed_cert, rl_cert = timed_leg()
        if not (torch.equal(exact_final, r_i32["score_final"]) and torch.equal(exact_ops, r_ops)):
            raise RuntimeError("strand-by-certificate leg produced different alignments than the exact leg")

    # further legs: the same batch split over `lanes_leg` chunks in flight (own stream + host thread each)
    elapsed_lanes = [0.0, 0.0]
    if args.lanes_leg > 1:
        ctx.set_lanes(args.lanes_leg)
        for which, exact in ((0, 1), (1, 0)):
            if exact == 0 and not args.certificate_leg:
                continue
            job.strand_by_certificate = 0 if exact else 1
            elapsed_lanes[which], _ = timed_leg()
            if args.certificate_leg and not (torch.equal(exact_final, r_i32["score_final"]) and torch.equal(exact_ops, r_ops)):
                raise RuntimeError("the lanes leg produced different alignments than the headline leg")
        ctx.set_lanes(max(1, args.lanes))
    job.strand_by_certificate = 0

    # ---- the dominant kernel on a device of its own (option sweeps_alone: the voted strand's chain finishes before the full sweeps start,
    # its prefix cells go to the front timer): what `roofline.dominant_kernel_alone_frac` is measured on.  Not part of any timed leg. ----
    rl_alone = None
    if args.alone_steps > 0:
        ctx.set_option("sweeps_alone", 1)
        try:
            for _ in range(2):
                run_call()
            lib.tracyhip_timing_enable(ctx._h, 1)
            lib.tracyhip_timing_reset(ctx._h)
            for _ in range(args.alone_steps):
                run_call()
            torch.cuda.synchronize()
            lib.tracyhip_timing_enable(ctx._h, 0)
            rl_alone = read_timers()
        finally:
            ctx.set_option("sweeps_alone", 0)

    if dist is not None:
        step()  # (the legs above ran other modes: one more headline step, untimed, whose gather is checked below)

    # ---- work done: DP cells of the four Gotoh calls per trace ----
    mt = mf - 2 * TRIM
    cells_rank = int(3 * mt * n * nt + (mf * slice_len).sum())
    tm = torch.tensor([elapsed, float(cells_rank), elapsed_cert, elapsed_lanes[0], elapsed_lanes[1], float(gathered["bytes"])], dtype=torch.float64,
                      device="cpu" if args.share_device else dev)
    elapsed_min = elapsed
    gathered_bytes_all = float(gathered["bytes"])
    if dist is not None:
        tmax = tm.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tm.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        tmin = tm.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        elapsed_max, cells_all, elapsed_cert_max = float(tmax[0]), float(tsum[1]), float(tmax[2])
        elapsed_lanes = [float(tmax[3]), float(tmax[4])]
        elapsed_min = float(tmin[0])
        gathered_bytes_all = float(tsum[5])
    else:
        elapsed_max, cells_all, elapsed_cert_max = elapsed, float(cells_rank), elapsed_cert

    # what rank 0 holds after the step's gather must be what the ranks computed: its own block, bit for bit (the other ranks' blocks are
    # compared with the single-process results in tests/test_shard_gloo.py and tests/test_gpu_bench_ranks.py)
    gather_ok = None
    if rank == 0 and gathered["out"] is not None and gathered["out"][0] is not None:
        allrec, got = gathered["out"]
        mine = torch.stack([r_i32[k] for k in REC_KEYS] + [r_fwd.to(torch.int32)], dim=1)
        gather_ok = gatherer.check_own_block(allrec, got, mine, [(r_ops, ops_cap, REC_KEYS.index("ops_len"))])

    # the headline's inputs and results are no longer needed on the device (the parity sample below reads host copies)
    sf_host, ol_host, ops_host = r_i32["score_final"].cpu().numpy(), r_i32["ops_len"].cpu().numpy(), None
    if rank == 0 and args.cpu_sample != 0:
        ops_host = r_ops.cpu().numpy()
    gathered["out"] = None
    if args.workload == "all":
        ctx.close()
        del d_refs, d_profs, r_ops, gatherer
        torch.cuda.empty_cache()
        for which in ("decompose", "allpairs", "seedextend", "cli"):
            extra[which] = run_extra(which)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    gcups = cells_all * args.steps / elapsed_max / 1e9
    tr, sc, bd, og = rl["trace"], rl["score"], rl["band"], rl["origin"]
    steps = max(args.steps, 1)

    def gbs(x):
        return x["bytes"] / (x["ms"] * 1e-3) / 1e9 if x["ms"] > 0 else 0.0

    def kgcups(x):
        return x["cells"] / (x["ms"] * 1e-3) / 1e9 if x["ms"] > 0 else 0.0
    # Dominant kernel of the step = the score-only Gotoh sweeps (DESIGN.md section 3 says what the launch holds and why the ceiling is VALU
    # issue: a score-only DP with resident inputs moves almost no bytes).  achieved / peak / frac are lane-operations per second; the HBM
    # figures stay beside them as hbm_*.  traffic = HBM bytes of one step's sweep launches from the committed rocprofv3 PMC passes.
    score_launch_ms = sc["ms"] / max(sc["launches"], 1)
    ops_per_cell = 8.0  # 16-bit formulation with the shared gap-open term: 4 v_add_u16 + 4 v_max_i16 per cell
    traffic, traffic_src, band_traffic, band_traffic_src = None, None, None, None
    try:
        from tools.legs import pmc_traffic
        traffic, traffic_src = pmc_traffic([r"gotoh_ckpt_prefix_kernel<"], "r[0-9][0-9]_pmc_hbm.json")
        band_traffic, band_traffic_src = pmc_traffic([r"band16_kernel<\d+, 0>", r"band16_multi3?(_counted)?_kernel<0>"], "r[0-9][0-9]_pmc_hbm.json")
    except Exception:  # noqa: BLE001
        pass
    clock_ghz, clock_src = None, None
    try:  # the clock the part holds under this kernel (it throttles below the 2.4 GHz the peak is priced at); from the committed profile
        from tools.legs import pmc_clock
        clock_ghz, clock_src = pmc_clock(r"gotoh_ckpt_prefix_kernel<15, 16, true", 1000000)
    except Exception:  # noqa: BLE001
        pass
    valu_achieved = kgcups(sc) * ops_per_cell / 1e3
    alone_frac = alone_ms = None
    if rl_alone is not None and rl_alone["score"]["ms"] > 0:
        alone_ms = rl_alone["score"]["ms"] / max(rl_alone["score"]["launches"], 1)
        alone_frac = kgcups(rl_alone["score"]) * ops_per_cell / 1e3 / 78.6
    short = lambda x: None if x is None else str(x).split(" ")[0]  # noqa: E731  (a profile file's name without the explanation)
    roofline = {"bound": "valu", "kernel": "gotoh_ckpt_prefix_kernel<15,16,compact,8>: 16-bit score-only Gotoh sweeps (DESIGN.md 2.2, 3)",
                "achieved": round(valu_achieved, 2), "peak": 78.6, "unit": "T lane-ops/s", "frac": round(valu_achieved / 78.6, 3),
                "ops_per_cell": ops_per_cell, "traffic": traffic, "traffic_source": short(traffic_src),
                "avg_launch_ms": round(score_launch_ms, 3), "launches": sc["launches"],
                "algorithmic_bytes_per_launch": sc["bytes"] // max(sc["launches"], 1), "kernel_gcups": round(kgcups(sc), 1),
                "share_of_step": round(sc["ms"] / steps / (elapsed_max / steps * 1e3), 3),
                "clock_ghz": clock_ghz, "clock_source": short(clock_src),
                "frac_at_measured_clock": None if not clock_ghz else round(valu_achieved / (78.6 * clock_ghz / 2.4), 3),
                "dominant_kernel_alone_ms": None if alone_ms is None else round(alone_ms, 3),
                "dominant_kernel_alone_frac": None if alone_frac is None else round(alone_frac, 3),
                "hbm_achieved_gbs": round(gbs(sc), 2), "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_frac": round(gbs(sc) / HBM_PEAK_GBS, 5),
                # the final alignments: tracebacks on certified diagonal bands (band16.h); cells / bytes are the bands'
                "traceback_hbm_gbs": round(gbs(tr), 1), "traceback_hbm_frac": round(gbs(tr) / HBM_PEAK_GBS, 4), "traceback_gcups": round(kgcups(tr), 1),
                "traceback_launch_ms": round(tr["ms"] / max(tr["launches"], 1), 3),
                "traceback_algorithmic_bytes": tr["bytes"] // max(tr["launches"], 1), "traceback_traffic": band_traffic,
                "ms_score": round(sc["ms"] / steps, 3), "ms_pruned_sweep_band": round(rl["front"]["ms"] / steps, 3),
                "ms_preliminary_ends": round(og["ms"] / steps, 3), "ms_final_alignments": round(tr["ms"] / steps, 3)}

    swept = sum(rl[k]["cells"] for k in ("score", "trace", "band", "prefix", "origin", "front")) / steps * world / (elapsed_max / steps) / 1e9
    line = {
        "metric": "GCUPS (tracy align: Gotoh affine-gap DP cells per second, whole job)",
        "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "backend": backend, "rccl_ranks": world if backend == "nccl" else 0, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int16 (score sweeps) / int32 (tracebacks)", "data": "synthetic",
        "traces_per_s": round(nt * world * args.steps / elapsed_max, 1),
        # `value` counts the reference's DP cells (3 x mt x n + mf x slice per trace); gcups_swept_cells prices the same step
        # by the cells the kernels really evaluated
        "gcups_swept_cells": round(swept, 2),
        "gathered_bytes_per_step": int(gathered_bytes_all), "gather_checked": gather_ok,
        # the part of ms_per_step spent behind the library call: records stacked, traceback strings packed (one device pass + its
        # synchronisation), both shipped to rank 0 (round 5's step gathered four integers per trace and nothing else)
        "gather_ms_per_step": round(gather_ms, 3),
        "config": {"workload": "configs[1]: %d synthetic %d-base traces `align` vs %d-base windows per GPU, scoring 3/-5/-10/-4, trims 50/50" % (nt, mf, n),
                   "traces_per_gpu": nt, "trace_len": mf, "ref_len": n,
                   "parallelism": "batch-sharded x%d, no data-path collective; gather of records + traceback strings to rank 0" % world,
                   "lanes_per_gpu": max(1, args.lanes), "gcups_swept_cells": round(swept, 2),
                   "gathered_bytes_per_step": int(gathered_bytes_all), "gather_ms_per_step": round(gather_ms, 3)},
        "roofline": roofline,
        # rank 0's view of a call: planned on the device, one host synchronisation (stream.hip); min / max over the ranks
        "pipeline": {"stream_ordered": call_stats["stream_ordered"], "host_syncs_per_call": call_stats["host_syncs"],
                     "traces_to_host_planned_tiers": call_stats["fallback_traces"], "ms_per_step_rank_min": round(elapsed_min / args.steps * 1e3, 3),
                     "ms_per_step_rank_max": round(elapsed_max / args.steps * 1e3, 3), "host_threads_per_rank": max(1, usable_cores() // max(1, world))},
    }
    if rl_cert is not None:
        # strand by certificate, timed on the same batch right after the headline leg; alignments checked identical above.
        # Reported for information: the loser's exact score is not computed, so it is not `value`.
        line["strand_by_certificate"] = {
            "ms_per_step": round(elapsed_cert_max / args.steps * 1e3, 3),
            "traces_per_s": round(nt * world * args.steps / elapsed_cert_max, 1),
            "cells_swept_fraction": round(sum(rl_cert[k]["cells"] for k in ("score", "prefix", "trace", "band", "origin", "front")) / steps / max(cells_rank, 1), 3),
            "alignments_identical_to_headline_leg": True,
        }
    if args.lanes_leg > 1 and elapsed_lanes[0] > 0:
        line["lanes"] = {"lanes": args.lanes_leg, "ms_per_step": round(elapsed_lanes[0] / args.steps * 1e3, 3),
                         "strand_by_certificate_ms_per_step": round(elapsed_lanes[1] / args.steps * 1e3, 3) if elapsed_lanes[1] > 0 else None}
    if True:  # (rank 0 at any N, on its share of the node's cores: every rank of a one-node job runs host work at the same time)
        nthreads = max(1, usable_cores() // max(1, world))  # one trace per thread
        sample = args.cpu_sample if args.cpu_sample >= 0 else max(8, min(40 * nthreads, 2048))  # ~10 s of CPU work
        if sample > 0:
            v, ns, dt, ores = cpu_baseline(profs, refs, nthreads, sample)
            v1, n1, dt1, _ = cpu_baseline(profs, refs, 1, 2)  # what one `tracy` process achieves
            model = ""
            try:
                model = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
            except (OSError, StopIteration):
                pass
            line["cpu_baseline"] = {"value": round(v, 4), "unit": "GCUPS", "cores": min(nthreads, ns), "kind": "port",
                                    "sample": "%d of the same traces through the oracle's sage.h chain (C, one trace per thread), %.1f s" % (ns, dt),
                                    "single_thread_gcups": round(v1, 4), "cpu_model": model, "usable_cores": nthreads}
            # the same traces must come out bit-identical on the GPU
            sf, ol, ops = sf_host, ol_host, ops_host
            ok = all(int(sf[i]) == o["score_final"] and ops[i * ops_cap:i * ops_cap + int(ol[i])].tobytes() == o["btr"]
                     for i, o in enumerate(ores))
            line["parity_checked"] = {"traces": ns, "bit_identical": bool(ok)}
    for k, v in extra.items():
        if v is not None:
            line[k] = v
    if args.share_device:
        line["shared_device"] = True  # (a test of the N-rank code path on one GPU: not a measurement)
        line["backend"] = "gloo"
    line = finish_line(line, args)
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
