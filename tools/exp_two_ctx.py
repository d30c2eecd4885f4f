#!/usr/bin/env python
"""experiment: one batch of `tracy align` traces through NCTX contexts (own stream + host thread each)"""
import argparse, ctypes as C, os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tracy_amd
from tracy_amd import capi, hostlib

ap = argparse.ArgumentParser()
ap.add_argument("--traces", type=int, default=10000)
ap.add_argument("--nctx", type=int, default=2)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--exact", type=int, default=1)
args = ap.parse_args()
nt, n, mf = args.traces, 10000, 1000
refs, profs, rev = hostlib.synth_align(1000, nt, n, mf, 0)
dev = torch.device("cuda", 0)
d_refs = torch.from_numpy(refs).cuda(); d_profs = torch.from_numpy(profs).cuda()
lib = capi.lib()
parts = []
per = nt // args.nctx
for c in range(args.nctx):
    lo, hi = c * per, (nt if c == args.nctx - 1 else (c + 1) * per)
    k = hi - lo
    pp_off = (np.arange(lo, hi, dtype=np.uint64) * np.uint64(6 * mf)); pp_len = np.full(k, mf, dtype=np.uint32)
    rr_off = (np.arange(lo, hi, dtype=np.uint64) * np.uint64(n)); rr_len = np.full(k, n, dtype=np.uint32)
    ops_cap = mf + n
    ops_off = (np.arange(k, dtype=np.uint64) * np.uint64(ops_cap))
    job = capi.AlignJob(); job.ntraces = k
    job.profiles = capi.SeqSet(capi.SEQ_PROFILE, d_profs.data_ptr(), pp_off.ctypes.data_as(C.POINTER(C.c_uint64)), pp_len.ctypes.data_as(C.POINTER(C.c_uint32)), k)
    job.refs = capi.SeqSet(capi.SEQ_CHAR, d_refs.data_ptr(), rr_off.ctypes.data_as(C.POINTER(C.c_uint64)), rr_len.ctypes.data_as(C.POINTER(C.c_uint32)), k)
    job.trim_left = 50; job.trim_right = 50; job.exact_orientation_scores = args.exact
    r_i32 = {kk: torch.zeros(k, dtype=torch.int32, device=dev) for kk in ("score_fwd", "score_rev", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final", "ops_len")}
    r_fwd = torch.zeros(k, dtype=torch.uint8, device=dev); r_ops = torch.zeros(k * ops_cap, dtype=torch.uint8, device=dev)
    out = capi.AlignResult()
    for kk, v in r_i32.items(): setattr(out, kk, v.data_ptr())
    out.forward = r_fwd.data_ptr(); out.ops = r_ops.data_ptr(); out.ops_offset = ops_off.ctypes.data_as(C.POINTER(C.c_uint64))
    ctx = tracy_amd.Context(0)
    st = torch.cuda.Stream()
    ctx.set_stream(st.cuda_stream)
    parts.append(dict(job=job, out=out, ctx=ctx, keep=(pp_off, pp_len, rr_off, rr_len, ops_off, r_i32, r_fwd, r_ops, st)))
prm = capi.Params(3, -5, -10, -4, 1, 0)

def run(p):
    rc = lib.tracyhip_align_traces(p["ctx"]._h, C.byref(p["job"]), C.byref(prm), capi.MEM_DEVICE, C.byref(p["out"]))
    assert rc == 0, lib.tracyhip_last_error()

def step():
    th = [threading.Thread(target=run, args=(p,)) for p in parts]
    for t in th: t.start()
    for t in th: t.join()

for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(args.steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
print("nctx %d exact %d: %.2f ms per step of %d traces" % (args.nctx, args.exact, dt * 1e3, nt))
