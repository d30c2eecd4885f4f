#!/usr/bin/env python
"""experiment: tracyhip_set_lanes sweep with / without kernel timers, default vs side stream"""
import argparse, ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tracy_amd
from tracy_amd import capi, hostlib
ap = argparse.ArgumentParser()
ap.add_argument("--traces", type=int, default=10000)
ap.add_argument("--steps", type=int, default=8)
args = ap.parse_args()
nt, n, mf = args.traces, 10000, 1000
refs, profs, rev = hostlib.synth_align(1000, nt, n, mf, 0)
dev = torch.device("cuda", 0)
d_refs = torch.from_numpy(refs).cuda(); d_profs = torch.from_numpy(profs).cuda()
lib = capi.lib()
pp_off = (np.arange(nt, dtype=np.uint64) * np.uint64(6 * mf)); pp_len = np.full(nt, mf, dtype=np.uint32)
rr_off = (np.arange(nt, dtype=np.uint64) * np.uint64(n)); rr_len = np.full(nt, n, dtype=np.uint32)
ops_cap = mf + n
ops_off = (np.arange(nt, dtype=np.uint64) * np.uint64(ops_cap))
job = capi.AlignJob(); job.ntraces = nt
job.profiles = capi.SeqSet(capi.SEQ_PROFILE, d_profs.data_ptr(), pp_off.ctypes.data_as(C.POINTER(C.c_uint64)), pp_len.ctypes.data_as(C.POINTER(C.c_uint32)), nt)
job.refs = capi.SeqSet(capi.SEQ_CHAR, d_refs.data_ptr(), rr_off.ctypes.data_as(C.POINTER(C.c_uint64)), rr_len.ctypes.data_as(C.POINTER(C.c_uint32)), nt)
job.trim_left = 50; job.trim_right = 50
r_i32 = {kk: torch.zeros(nt, dtype=torch.int32, device=dev) for kk in ("score_fwd", "score_rev", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final", "ops_len")}
r_fwd = torch.zeros(nt, dtype=torch.uint8, device=dev); r_ops = torch.zeros(nt * ops_cap, dtype=torch.uint8, device=dev)
out = capi.AlignResult()
for kk, v in r_i32.items(): setattr(out, kk, v.data_ptr())
out.forward = r_fwd.data_ptr(); out.ops = r_ops.data_ptr(); out.ops_offset = ops_off.ctypes.data_as(C.POINTER(C.c_uint64))
prm = capi.Params(3, -5, -10, -4, 1, 0)
ctx = tracy_amd.Context(0)
side = torch.cuda.Stream()
for stream_kind in ("default", "side"):
    ctx.set_stream(torch.cuda.current_stream().cuda_stream if stream_kind == "default" else side.cuda_stream)
    for timing in (0, 1):
        for lanes in (1, 2, 3, 4):
            ctx.set_lanes(lanes)
            lib.tracyhip_timing_enable(ctx._h, timing)
            for exact in (1, 0):
                job.strand_by_certificate = 0 if exact else 1
                def step():
                    rc = lib.tracyhip_align_traces(ctx._h, C.byref(job), C.byref(prm), capi.MEM_DEVICE, C.byref(out))
                    assert rc == 0, lib.tracyhip_last_error()
                for _ in range(2): step()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(args.steps): step()
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
                print("stream %-7s timing %d lanes %d exact %d: %.2f ms" % (stream_kind, timing, lanes, exact, dt * 1e3), flush=True)
