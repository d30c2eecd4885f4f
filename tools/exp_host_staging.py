#!/usr/bin/env python
"""PCIe-inclusive rate of tracyhip_align_traces: the bench batch with TRACYHIP_MEM_HOST (pageable and pinned host buffers)"""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tracy_amd
from tracy_amd import capi, hostlib
nt, n, mf = 10000, 10000, 1000
refs, profs, rev = hostlib.synth_align(1000, nt, n, mf, 0)
lib = capi.lib()
pp_off = (np.arange(nt, dtype=np.uint64) * np.uint64(6 * mf)); pp_len = np.full(nt, mf, dtype=np.uint32)
rr_off = (np.arange(nt, dtype=np.uint64) * np.uint64(n)); rr_len = np.full(nt, n, dtype=np.uint32)
ops_cap = mf + n
ops_off = (np.arange(nt, dtype=np.uint64) * np.uint64(ops_cap))
prm = capi.Params(3, -5, -10, -4, 1, 0)
for pinned in (0, 1):
    if pinned:
        t_refs = torch.from_numpy(refs).pin_memory(); t_profs = torch.from_numpy(profs).pin_memory()
        mk = lambda shape, dt: torch.zeros(shape, dtype=dt).pin_memory()
    else:
        t_refs = torch.from_numpy(refs); t_profs = torch.from_numpy(profs)
        mk = lambda shape, dt: torch.zeros(shape, dtype=dt)
    job = capi.AlignJob(); job.ntraces = nt
    job.profiles = capi.SeqSet(capi.SEQ_PROFILE, t_profs.data_ptr(), pp_off.ctypes.data_as(C.POINTER(C.c_uint64)), pp_len.ctypes.data_as(C.POINTER(C.c_uint32)), nt)
    job.refs = capi.SeqSet(capi.SEQ_CHAR, t_refs.data_ptr(), rr_off.ctypes.data_as(C.POINTER(C.c_uint64)), rr_len.ctypes.data_as(C.POINTER(C.c_uint32)), nt)
    job.trim_left = 50; job.trim_right = 50
    r_i32 = {kk: mk(nt, torch.int32) for kk in ("score_fwd", "score_rev", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final", "ops_len")}
    r_fwd = mk(nt, torch.uint8); r_ops = mk(nt * ops_cap, torch.uint8)
    out = capi.AlignResult()
    for kk, v in r_i32.items(): setattr(out, kk, v.data_ptr())
    out.forward = r_fwd.data_ptr(); out.ops = r_ops.data_ptr(); out.ops_offset = ops_off.ctypes.data_as(C.POINTER(C.c_uint64))
    ctx = tracy_amd.Context(0)
    for lanes in (1, 2, 3):
        ctx.set_lanes(lanes)
        for exact in (1, 0):
            job.strand_by_certificate = 0 if exact else 1
            def step():
                rc = lib.tracyhip_align_traces(ctx._h, C.byref(job), C.byref(prm), capi.MEM_HOST, C.byref(out))
                assert rc == 0, lib.tracyhip_last_error()
            for _ in range(2): step()
            t0 = time.perf_counter()
            for _ in range(5): step()
            dt = (time.perf_counter() - t0) / 5
            print("MEM_HOST pinned %d lanes %d exact %d: %.2f ms per step (%.0f traces/s)" % (pinned, lanes, exact, dt * 1e3, nt / dt), flush=True)
    del ctx
