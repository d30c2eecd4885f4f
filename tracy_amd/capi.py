"""ctypes binding of include/tracy_hip.h (plumbing for tests and bench.py; plain pointers and sizes).

Loading fails loudly when the in-tree library is missing: build it with `python tracy_amd/build.py`
(or __graft_entry__.build()).  Nothing here computes on the CPU.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

OK, ERR_ARG, ERR_HIP, ERR_OOM, ERR_RANGE, ERR_NODEVICE = 0, -1, -2, -3, -4, -5
MEM_HOST, MEM_DEVICE = 0, 1
SEQ_CHAR, SEQ_PROFILE = 0, 1


class TracyHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tracyhip error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("go", C.c_int32), ("ge", C.c_int32),
                ("hfree", C.c_int32), ("vfree", C.c_int32)]


class SeqSet(C.Structure):
    _fields_ = [("kind", C.c_int32), ("data", C.c_void_p), ("offset", C.POINTER(C.c_uint64)),
                ("length", C.POINTER(C.c_uint32)), ("count", C.c_uint32)]


class Pairs(C.Structure):
    _fields_ = [("npairs", C.c_uint32), ("a1", SeqSet), ("a2", SeqSet), ("a1_index", C.POINTER(C.c_uint32)),
                ("a2_index", C.POINTER(C.c_uint32))]


class AlignJob(C.Structure):
    _fields_ = [("ntraces", C.c_uint32), ("profiles", SeqSet), ("refs", SeqSet), ("ref_index", C.POINTER(C.c_uint32)),
                ("trim_left", C.c_uint32), ("trim_right", C.c_uint32), ("oriented", C.POINTER(C.c_uint8)),
                ("strand_by_certificate", C.c_uint32)]


class AlignResult(C.Structure):
    _fields_ = [("score_fwd", C.c_void_p), ("score_rev", C.c_void_p), ("forward", C.c_void_p),
                ("score_prelim", C.c_void_p), ("slice_begin", C.c_void_p), ("slice_len", C.c_void_p),
                ("ref_pos", C.c_void_p), ("score_final", C.c_void_p), ("ops", C.c_void_p),
                ("ops_offset", C.POINTER(C.c_uint64)), ("ops_len", C.c_void_p)]


class CallStats(C.Structure):
    _fields_ = [("traces", C.c_uint32), ("stream_ordered", C.c_uint32), ("host_syncs", C.c_uint32), ("fallback_traces", C.c_uint32),
                ("pruned", C.c_uint32), ("pruned_uncertified", C.c_uint32), ("prelim_banded", C.c_uint32), ("prelim_repeated", C.c_uint32),
                ("final_banded", C.c_uint32), ("final_repeated", C.c_uint32), ("allele_pruned", C.c_uint32 * 2),
                ("allele_uncertified", C.c_uint32 * 2), ("allele_banded", C.c_uint32 * 3), ("allele_repeated", C.c_uint32 * 3),
                ("allele_shared_prefix", C.c_uint32)]


def library_path():
    return os.path.join(_HERE, "lib", "libtracy_hip.so")


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        p = library_path()
        if not os.path.exists(p):
            raise ImportError("tracy_amd: %s is missing -- run `python tracy_amd/build.py` (hipcc, gfx950). "
                              "There is no CPU fallback." % p)
        _LIB = C.CDLL(p)
        _LIB.tracyhip_last_error.restype = C.c_char_p
        _LIB.tracyhip_version.restype = C.c_char_p
        _LIB.tracyhip_group_context.restype = C.c_void_p
    return _LIB


def _check(rc):
    if rc != OK:
        raise TracyHipError(rc, lib().tracyhip_last_error().decode())


def device_count():
    n = C.c_int(0)
    rc = lib().tracyhip_device_count(C.byref(n))
    return n.value if rc == OK else 0


class PackedSeqs:
    """Host-side packing of a list of sequences (bytes) or profiles (float32 [6][len])."""

    def __init__(self, seqs, kind=None):
        if kind is None:
            kind = SEQ_CHAR if (len(seqs) == 0 or isinstance(seqs[0], (bytes, bytearray))) else SEQ_PROFILE
        self.kind = kind
        self.count = len(seqs)
        self.length = np.zeros(max(self.count, 1), dtype=np.uint32)
        self.offset = np.zeros(max(self.count, 1), dtype=np.uint64)
        pos = 0
        parts = []
        for i, s in enumerate(seqs):
            if kind == SEQ_CHAR:
                a = np.frombuffer(bytes(s), dtype=np.uint8)
                ln = a.size
            else:
                a = np.ascontiguousarray(s, dtype=np.float32)
                assert a.ndim == 2 and a.shape[0] == 6
                ln = a.shape[1]
                a = a.reshape(-1)
            self.length[i] = ln
            self.offset[i] = pos
            pos += a.size
            parts.append(a)
        dt = np.uint8 if kind == SEQ_CHAR else np.float32
        self.data = np.concatenate(parts) if parts and pos else np.zeros(1, dtype=dt)
        self.nelem = pos

    def seqset(self, data_ptr=None):
        s = SeqSet()
        s.kind = self.kind
        s.data = data_ptr if data_ptr is not None else self.data.ctypes.data
        s.offset = self.offset.ctypes.data_as(C.POINTER(C.c_uint64))
        s.length = self.length.ctypes.data_as(C.POINTER(C.c_uint32))
        s.count = self.count
        return s


class Context:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().tracyhip_create(int(device), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().tracyhip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(lib().tracyhip_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_workspace_limit(self, nbytes):
        _check(lib().tracyhip_set_workspace_limit(self._h, C.c_uint64(nbytes)))

    def set_lanes(self, n):
        """batch pipelines split a call into n chunks in flight (own stream + host thread each); 1 = off"""
        _check(lib().tracyhip_set_lanes(self._h, C.c_uint32(n)))

    def synchronize(self):
        _check(lib().tracyhip_synchronize(self._h))

    def set_option(self, name, value):
        """tracyhip_set_option: name without the TRACYHIP_ prefix (the environment is read once, at creation)"""
        _check(lib().tracyhip_set_option(self._h, str(name).encode(), str(int(value) if isinstance(value, bool) else value).encode()))

    def describe(self):
        n = lib().tracyhip_describe(self._h, None, C.c_size_t(0))
        buf = C.create_string_buffer(n)
        lib().tracyhip_describe(self._h, buf, C.c_size_t(n))
        return dict(line.split("=", 1) for line in buf.value.decode().splitlines())

    def last_call_stats(self):
        """tiers the traces of the last align_traces / decompose_traces call took (tracyhip_last_call_stats)"""
        st = CallStats()
        _check(lib().tracyhip_last_call_stats(self._h, C.byref(st)))
        out = {}
        for name, ty in CallStats._fields_:
            v = getattr(st, name)
            out[name] = list(v) if hasattr(v, "__len__") else int(v)
        return out

    # ---- host-buffer convenience wrappers (lists in, numpy out) -----------------------------------
    @staticmethod
    def _pairs(a1, a2, idx1=None, idx2=None):
        p1 = a1 if isinstance(a1, PackedSeqs) else PackedSeqs(a1)
        p2 = a2 if isinstance(a2, PackedSeqs) else PackedSeqs(a2)
        pr = Pairs()
        keep = [p1, p2]
        if idx1 is not None:
            idx1 = np.ascontiguousarray(idx1, dtype=np.uint32)
            idx2 = np.ascontiguousarray(idx2, dtype=np.uint32)
            pr.npairs = len(idx1)
            pr.a1_index = idx1.ctypes.data_as(C.POINTER(C.c_uint32))
            pr.a2_index = idx2.ctypes.data_as(C.POINTER(C.c_uint32))
            keep += [idx1, idx2]
        else:
            assert p1.count == p2.count
            pr.npairs = p1.count
        pr.a1 = p1.seqset()
        pr.a2 = p2.seqset()
        return pr, keep, p1, p2

    def _pair_lengths(self, pr, p1, p2, idx1, idx2):
        if idx1 is None:
            return p1.length[:pr.npairs].astype(np.uint64), p2.length[:pr.npairs].astype(np.uint64)
        return p1.length[np.asarray(idx1)].astype(np.uint64), p2.length[np.asarray(idx2)].astype(np.uint64)

    def score(self, a1, a2, params, needle=False, idx1=None, idx2=None):
        pr, keep, p1, p2 = self._pairs(a1, a2, idx1, idx2)
        prm = Params(*params)
        out = np.zeros(max(pr.npairs, 1), dtype=np.int32)
        fn = lib().tracyhip_needle_score if needle else lib().tracyhip_gotoh_score
        _check(fn(self._h, C.byref(pr), C.byref(prm), MEM_HOST, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out[:pr.npairs]

    def align(self, a1, a2, params, needle=False, idx1=None, idx2=None, rows=False):
        """returns (scores, [btr bytes per pair, push order]) and optionally the alignment rows"""
        pr, keep, p1, p2 = self._pairs(a1, a2, idx1, idx2)
        prm = Params(*params)
        n = pr.npairs
        l1, l2 = self._pair_lengths(pr, p1, p2, idx1, idx2)
        cap = (l1 + l2).astype(np.uint64)
        off = np.zeros(max(n, 1), dtype=np.uint64)
        if n:
            off[1:n] = np.cumsum(cap)[:-1]
        total = int(cap.sum()) if n else 0
        ops = np.zeros(max(total, 1), dtype=np.uint8)
        olen = np.zeros(max(n, 1), dtype=np.uint32)
        scores = np.zeros(max(n, 1), dtype=np.int32)
        fn = lib().tracyhip_needle_align if needle else lib().tracyhip_gotoh_align
        _check(fn(self._h, C.byref(pr), C.byref(prm), MEM_HOST, scores.ctypes.data_as(C.POINTER(C.c_int32)),
                  ops.ctypes.data_as(C.POINTER(C.c_uint8)), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                  olen.ctypes.data_as(C.POINTER(C.c_uint32))))
        btr = [ops[int(off[i]):int(off[i]) + int(olen[i])].tobytes() for i in range(n)]
        if not rows:
            return scores[:n], btr
        r0 = np.zeros(max(total, 1), dtype=np.uint8)
        r1 = np.zeros(max(total, 1), dtype=np.uint8)
        _check(lib().tracyhip_alignment_rows(self._h, C.byref(pr), MEM_HOST, ops.ctypes.data_as(C.POINTER(C.c_uint8)),
                                             off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                             olen.ctypes.data_as(C.POINTER(C.c_uint32)),
                                             r0.ctypes.data_as(C.POINTER(C.c_uint8)),
                                             r1.ctypes.data_as(C.POINTER(C.c_uint8))))
        rws = [(r0[int(off[i]):int(off[i]) + int(olen[i])].tobytes(), r1[int(off[i]):int(off[i]) + int(olen[i])].tobytes())
               for i in range(n)]
        return scores[:n], btr, rws


def _align_banded(self, a1, a2, params, band_lo, band_hi, origin=False):
    """tracyhip_gotoh_banded with host buffers: (scores, btr list) or, origin=True, (scores, [(lead, c_e)])"""
    pr, keep, p1, p2 = self._pairs(a1, a2, None, None)
    prm = Params(*params)
    n = pr.npairs
    l1, l2 = self._pair_lengths(pr, p1, p2, None, None)
    cap = (l1 + l2).astype(np.uint64)
    off = np.zeros(max(n, 1), dtype=np.uint64)
    if n:
        off[1:n] = np.cumsum(cap)[:-1]
    ops = np.zeros(max(int(cap.sum()) if n else 0, 1), dtype=np.uint8)
    olen = np.zeros(max(n, 1), dtype=np.uint32)
    scores = np.zeros(max(n, 1), dtype=np.int32)
    ends = np.zeros(2 * max(n, 1), dtype=np.uint32)
    lo = np.ascontiguousarray(band_lo, dtype=np.int32)
    hi = np.ascontiguousarray(band_hi, dtype=np.int32)
    i32p = C.POINTER(C.c_int32)
    _check(lib().tracyhip_gotoh_banded(self._h, C.byref(pr), C.byref(prm), lo.ctypes.data_as(i32p), hi.ctypes.data_as(i32p), MEM_HOST,
                                       scores.ctypes.data_as(i32p), ops.ctypes.data_as(C.POINTER(C.c_uint8)), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       olen.ctypes.data_as(C.POINTER(C.c_uint32)), ends.ctypes.data_as(C.POINTER(C.c_uint32)) if origin else None))
    if origin:
        return scores[:n], [(int(ends[2 * i]), int(ends[2 * i + 1])) for i in range(n)]
    return scores[:n], [ops[int(off[i]):int(off[i]) + int(olen[i])].tobytes() for i in range(n)]


class PreparedAlign:
    """host-buffer job / result structs of tracyhip_align_traces (everything they point to is kept alive by this object)"""

    def __init__(self, profiles, refs, params, trim_left=50, trim_right=50, ref_index=None, oriented=None, exact_scores=True):
        pp = profiles if isinstance(profiles, PackedSeqs) else PackedSeqs(profiles, SEQ_PROFILE)
        pr = refs if isinstance(refs, PackedSeqs) else PackedSeqs(refs, SEQ_CHAR)
        nt = self.nt = pp.count
        job = self.job = AlignJob()
        job.ntraces = nt
        job.profiles = pp.seqset()
        job.refs = pr.seqset()
        self.keep = [pp, pr]
        if ref_index is not None:
            ref_index = np.ascontiguousarray(ref_index, dtype=np.uint32)
            job.ref_index = ref_index.ctypes.data_as(C.POINTER(C.c_uint32))
            rlen = pr.length[ref_index]
            self.keep.append(ref_index)
        else:
            rlen = pr.length[:nt]
        job.trim_left = trim_left
        job.trim_right = trim_right
        job.strand_by_certificate = 0 if exact_scores else 1  # opt-in: the losing strand may carry a certified upper bound
        if oriented is not None:
            oriented = np.ascontiguousarray(oriented, dtype=np.uint8)
            job.oriented = oriented.ctypes.data_as(C.POINTER(C.c_uint8))
            self.keep.append(oriented)
        cap = pp.length[:nt].astype(np.uint64) + rlen.astype(np.uint64)
        off = self.off = np.zeros(max(nt, 1), dtype=np.uint64)
        if nt:
            off[1:nt] = np.cumsum(cap)[:-1]
        res = self.res = {
            "score_fwd": np.zeros(max(nt, 1), np.int32), "score_rev": np.zeros(max(nt, 1), np.int32),
            "forward": np.zeros(max(nt, 1), np.uint8), "score_prelim": np.zeros(max(nt, 1), np.int32),
            "slice_begin": np.zeros(max(nt, 1), np.uint32), "slice_len": np.zeros(max(nt, 1), np.uint32),
            "ref_pos": np.zeros(max(nt, 1), np.uint32), "score_final": np.zeros(max(nt, 1), np.int32),
            "ops": np.zeros(max(int(cap.sum()) if nt else 0, 1), np.uint8), "ops_len": np.zeros(max(nt, 1), np.uint32),
        }
        out = self.out = AlignResult()
        for k in ("score_fwd", "score_rev", "forward", "score_prelim", "slice_begin", "slice_len", "ref_pos",
                  "score_final", "ops", "ops_len"):
            setattr(out, k, res[k].ctypes.data)
        out.ops_offset = off.ctypes.data_as(C.POINTER(C.c_uint64))
        self.prm = Params(params[0], params[1], params[2], params[3], 1, 0)

    def results(self):
        res, off, nt = dict(self.res), self.off, self.nt
        btr = [res["ops"][int(off[i]):int(off[i]) + int(res["ops_len"][i])].tobytes() for i in range(nt)]
        for k in list(res):
            if k != "ops":
                res[k] = res[k][:nt]
        res["btr"] = btr
        return res


def _align_traces(self, profiles, refs, params, trim_left=50, trim_right=50, ref_index=None, oriented=None, exact_scores=True, device=False):
    """tracyhip_align_traces with host buffers.  profiles: list of float32 [6][mf]; refs: list of bytes.
    oriented: None, or rs.forward per trace when the references are already oriented (indexed-genome path).
    device=True: payloads and result arrays in device memory (torch tensors, TRACYHIP_MEM_DEVICE), copied back afterwards.
    Returns a dict of numpy arrays + the list of final traceback strings (push order)."""
    p = PreparedAlign(profiles, refs, params, trim_left, trim_right, ref_index, oriented, exact_scores)
    if device:
        import torch
        pp, pr = p.keep[0], p.keep[1]
        dp, dr = torch.from_numpy(pp.data).cuda(), torch.from_numpy(pr.data).cuda()
        p.job.profiles = pp.seqset(dp.data_ptr())
        p.job.refs = pr.seqset(dr.data_ptr())
        dres = {k: torch.zeros(v.shape, dtype=getattr(torch, str(v.dtype)) if str(v.dtype) != "uint32" else torch.int32, device="cuda") for k, v in p.res.items()}
        for k, v in dres.items():
            setattr(p.out, k, v.data_ptr())
        torch.cuda.synchronize()
        _check(lib().tracyhip_align_traces(self._h, C.byref(p.job), C.byref(p.prm), MEM_DEVICE, C.byref(p.out)))
        torch.cuda.synchronize()
        for k, v in dres.items():
            p.res[k] = v.cpu().numpy().view(p.res[k].dtype)
        return p.results()
    if isinstance(self, Group):
        _check(lib().tracyhip_group_align_traces(self._g, C.byref(p.job), C.byref(p.prm), C.byref(p.out)))
    else:
        _check(lib().tracyhip_align_traces(self._h, C.byref(p.job), C.byref(p.prm), MEM_HOST, C.byref(p.out)))
    return p.results()


Context.align_traces = _align_traces


def _align_traces_async(self, job, prm, out, mem=MEM_DEVICE):
    """tracyhip_align_traces_async on prepared structs (they, and everything they point to, must outlive synchronize())"""
    _check(lib().tracyhip_align_traces_async(self._h, C.byref(job), C.byref(prm), mem, C.byref(out)))


Context.align_traces_async = _align_traces_async


class RaggedSrc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("stride_bytes", C.c_uint64), ("elem_bytes", C.c_uint32), ("lens", C.c_void_p), ("lens_stride", C.c_uint32)]


def _pack_ragged_multi(self, kinds, n, out=None):
    """tracyhip_pack_ragged_multi on torch CUDA tensors.  kinds: [(src, stride in ELEMENTS, lens tensor (uint32 / int32), lens_stride)]:
    region i of a kind = `stride` elements of src from i * stride, of which the first lens[i * lens_stride] are used.  Returns (packed uint8
    tensor -- kind-major --, [bytes per kind]); `out`: a uint8 tensor of at least sum(n * stride * itemsize) bytes to pack into (kept by a
    caller that packs every step)."""
    import torch
    arr = (RaggedSrc * len(kinds))()
    cap = 0
    for k, (src, stride, lens, lens_stride) in enumerate(kinds):
        elem = src.element_size()
        arr[k] = RaggedSrc(src.data_ptr(), int(stride) * elem, elem, lens.data_ptr(), int(lens_stride))
        cap += int(n) * int(stride) * elem
    if out is None or out.numel() < cap:
        out = torch.empty(max(cap, 1), dtype=torch.uint8, device=kinds[0][0].device)
    kb = (C.c_uint64 * len(kinds))()
    _check(lib().tracyhip_pack_ragged_multi(self._h, arr, C.c_uint32(len(kinds)), C.c_uint32(int(n)), C.c_void_p(out.data_ptr()), C.c_uint64(out.numel()), kb))
    sizes = [int(x) for x in kb]
    return out[:sum(sizes)], sizes


def _pack_ragged(self, src, stride, lens, n=None, lens_stride=1, out=None):
    """one payload kind (tracyhip_pack_ragged): returns (packed uint8 tensor, bytes)"""
    if n is None:
        n = int(lens.numel()) // lens_stride
    packed, sizes = _pack_ragged_multi(self, [(src, stride, lens, lens_stride)], n, out)
    return packed, sizes[0]


Context.pack_ragged = _pack_ragged
Context.pack_ragged_multi = _pack_ragged_multi


def pair_bounds(len1, len2, idx1, idx2, parts):
    """tracyhip_pair_bounds: boundaries of `parts` contiguous slices of a pair list with (nearly) equal DP cell count.
    Pure host arithmetic inside the library (works without a GPU) -- the one rule used by device groups and by rank sharding."""
    len1 = np.ascontiguousarray(len1, dtype=np.uint32)
    len2 = np.ascontiguousarray(len2, dtype=np.uint32)
    idx1 = np.ascontiguousarray(idx1, dtype=np.uint32)
    idx2 = np.ascontiguousarray(idx2, dtype=np.uint32)
    pr = Pairs()
    pr.npairs = len(idx1)
    pr.a1 = SeqSet(SEQ_CHAR, None, None, len1.ctypes.data_as(C.POINTER(C.c_uint32)), len(len1))
    pr.a2 = SeqSet(SEQ_CHAR, None, None, len2.ctypes.data_as(C.POINTER(C.c_uint32)), len(len2))
    pr.a1_index = idx1.ctypes.data_as(C.POINTER(C.c_uint32))
    pr.a2_index = idx2.ctypes.data_as(C.POINTER(C.c_uint32))
    b = np.zeros(parts + 1, dtype=np.uint64)
    _check(lib().tracyhip_pair_bounds(C.byref(pr), C.c_uint32(parts), b.ctypes.data_as(C.POINTER(C.c_uint64))))
    return b.astype(np.int64)


class Group:
    """tracyhip_group: one context per listed device (a device may repeat), host buffers, one host thread per member"""

    def __init__(self, devices=None, ndevices=0):
        self._g = C.c_void_p()
        if devices is None:
            _check(lib().tracyhip_group_create(None, int(ndevices), C.byref(self._g)))
        else:
            arr = (C.c_int * len(devices))(*devices)
            _check(lib().tracyhip_group_create(arr, len(devices), C.byref(self._g)))

    def close(self):
        if self._g:
            lib().tracyhip_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return lib().tracyhip_group_size(self._g)

    def set_lanes(self, n):
        _check(lib().tracyhip_group_set_lanes(self._g, C.c_uint32(n)))

    def score(self, a1, a2, params, idx1=None, idx2=None):
        pr, keep, p1, p2 = Context._pairs(a1, a2, idx1, idx2)
        prm = Params(*params)
        out = np.zeros(max(pr.npairs, 1), dtype=np.int32)
        _check(lib().tracyhip_group_gotoh_score(self._g, C.byref(pr), C.byref(prm), out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out[:pr.npairs]


class KernelTiming(C.Structure):
    _fields_ = [("ms", C.c_double), ("launches", C.c_uint64), ("cells", C.c_uint64), ("bytes", C.c_uint64)]


# ---- allele deconvolution --------------------------------------------------------------------------
class Breakpoint(C.Structure):
    _fields_ = [("indelshift", C.c_int32), ("traceleft", C.c_int32), ("breakpoint", C.c_uint32), ("best_diff", C.c_float)]


class BaseCallsBatch(C.Structure):
    _fields_ = [("ntraces", C.c_uint32), ("signal", C.c_void_p), ("signal_offset", C.POINTER(C.c_uint64)),
                ("nsamples", C.POINTER(C.c_uint32)), ("bcpos", C.c_void_p), ("primary", C.c_void_p),
                ("secondary", C.c_void_p), ("bc_offset", C.POINTER(C.c_uint64)), ("bc_len", C.POINTER(C.c_uint32)),
                ("peaks", C.c_void_p)]  # (optional: the four channels at every basecall's peak position; None = built on the device)


class DecompParams(C.Structure):
    _fields_ = [("trim_left", C.c_int32), ("trim_right", C.c_int32), ("maxindel", C.c_int32), ("madc", C.c_int32)]


class DecompStatus(C.Structure):
    _fields_ = [("kind", C.c_int32), ("best_ins", C.c_int32), ("best_del", C.c_int32), ("best_fr", C.c_int32),
                ("dcp_n", C.c_uint32), ("pad", C.c_uint32)]


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class HostBaseCalls:
    """pack lists of (signal [4][ns] int32, bcpos, primary, secondary) for the host-buffer entry points"""

    def __init__(self, signals, bcpos, primary, secondary):
        n = len(signals)
        self.n = n
        self.nsamples = np.array([s.shape[1] for s in signals], dtype=np.uint32)
        self.sig_off = np.zeros(max(n, 1), dtype=np.uint64)
        self.bc_len = np.array([len(p) for p in primary], dtype=np.uint32)
        self.bc_off = np.zeros(max(n, 1), dtype=np.uint64)
        so, bo = 0, 0
        for i in range(n):
            self.sig_off[i] = so
            so += 4 * int(self.nsamples[i])
            self.bc_off[i] = bo
            bo += int(self.bc_len[i])
        self.signal = np.concatenate([np.ascontiguousarray(s, dtype=np.int32).reshape(-1) for s in signals]) if n else np.zeros(1, np.int32)
        self.bcpos = np.concatenate([np.ascontiguousarray(b, dtype=np.int32) for b in bcpos]) if n else np.zeros(1, np.int32)
        self.primary = np.frombuffer(b"".join(bytes(p) for p in primary), dtype=np.uint8).copy() if n else np.zeros(1, np.uint8)
        self.secondary = np.frombuffer(b"".join(bytes(p) for p in secondary), dtype=np.uint8).copy() if n else np.zeros(1, np.uint8)

    def peak_table(self):
        """tracyhip_basecalls::peaks of the batch: int32 [sum of bc_len][4] = the four channels at every basecall's peak position"""
        out = np.zeros((max(int(self.bc_len.sum()), 1), 4), dtype=np.int32)
        for i in range(self.n):
            so, ns, bo, bl = int(self.sig_off[i]), int(self.nsamples[i]), int(self.bc_off[i]), int(self.bc_len[i])
            sig = self.signal[so:so + 4 * ns].reshape(4, ns)
            out[bo:bo + bl] = sig[:, self.bcpos[bo:bo + bl]].T
        return out

    def struct(self, peaks_only=False):
        """peaks_only: the peak table instead of the chromatograms (signal / bcpos NULL)"""
        b = BaseCallsBatch()
        b.ntraces = self.n
        if peaks_only:
            self._peaks = self.peak_table()
            b.peaks = self._peaks.ctypes.data
            b.primary = self.primary.ctypes.data
            b.secondary = self.secondary.ctypes.data
            b.bc_offset = _u64p(self.bc_off)
            b.bc_len = _u32p(self.bc_len)
            return b
        b.signal = self.signal.ctypes.data
        b.signal_offset = _u64p(self.sig_off)
        b.nsamples = _u32p(self.nsamples)
        b.bcpos = self.bcpos.ctypes.data
        b.primary = self.primary.ctypes.data
        b.secondary = self.secondary.ctypes.data
        b.bc_offset = _u64p(self.bc_off)
        b.bc_len = _u32p(self.bc_len)
        return b

    def split(self, arr):
        return [arr[int(self.bc_off[i]):int(self.bc_off[i]) + int(self.bc_len[i])].tobytes() for i in range(self.n)]


def _find_breakpoint(self, profiles):
    pp = profiles if isinstance(profiles, PackedSeqs) else PackedSeqs(profiles, SEQ_PROFILE)
    out = (Breakpoint * max(pp.count, 1))()
    ss = pp.seqset()
    _check(lib().tracyhip_find_breakpoint(self._h, C.byref(ss), MEM_HOST, out))
    return [out[i] for i in range(pp.count)]


def _pack_rows(rows):
    n = len(rows)
    lens = np.array([len(r[0]) for r in rows], dtype=np.uint32)
    off = np.zeros(max(n, 1), dtype=np.uint64)
    if n:
        off[1:n] = np.cumsum(lens.astype(np.uint64))[:-1]
    r0 = np.frombuffer(b"".join(r[0] for r in rows) + b"\0", dtype=np.uint8).copy()
    r1 = np.frombuffer(b"".join(r[1] for r in rows) + b"\0", dtype=np.uint8).copy()
    return r0, r1, off, lens


def _find_homozygous_breakpoint(self, rows, bps):
    n = len(rows)
    r0, r1, off, lens = _pack_rows(rows)
    arr = (Breakpoint * max(n, 1))(*bps)
    status = np.zeros(max(n, 1), dtype=np.int32)
    _check(lib().tracyhip_find_homozygous_breakpoint(self._h, n, r0.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                     r1.ctypes.data_as(C.POINTER(C.c_uint8)), _u64p(off), _u32p(lens), MEM_HOST,
                                                     arr, status.ctypes.data_as(C.POINTER(C.c_int32))))
    return [arr[i] for i in range(n)], status[:n]


def _decompose_alleles(self, hbc, rows, bps, refslice_len, trim_left=50, trim_right=50, maxindel=1000, madc=5):
    n = hbc.n
    r0, r1, off, lens = _pack_rows(rows)
    arr = (Breakpoint * max(n, 1))(*bps)
    rl = np.ascontiguousarray(refslice_len, dtype=np.uint32)
    cap = 2 * maxindel + 2
    doff = (np.arange(max(n, 1), dtype=np.uint64) * np.uint64(cap))
    di = np.zeros(max(n, 1) * cap, dtype=np.int32)
    de = np.zeros(max(n, 1) * cap, dtype=np.int32)
    st = (DecompStatus * max(n, 1))()
    prm = DecompParams(trim_left, trim_right, maxindel, madc)
    b = hbc.struct()
    _check(lib().tracyhip_decompose_alleles(self._h, C.byref(b), r0.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            r1.ctypes.data_as(C.POINTER(C.c_uint8)), _u64p(off), _u32p(lens), arr, _u32p(rl),
                                            C.byref(prm), MEM_HOST, di.ctypes.data_as(C.POINTER(C.c_int32)),
                                            de.ctypes.data_as(C.POINTER(C.c_int32)), _u64p(doff), st))
    dcp = [[(int(di[i * cap + k]), int(de[i * cap + k])) for k in range(st[i].dcp_n)] for i in range(n)]
    status = [(st[i].kind, st[i].best_ins, st[i].best_del, st[i].best_fr) for i in range(n)]
    return hbc.split(hbc.primary), hbc.split(hbc.secondary), dcp, status


def _secondary_decomposed(self, hbc, peaks_only=False):
    out = np.zeros(max(len(hbc.primary), 1), dtype=np.uint8)
    b = hbc.struct(peaks_only)
    _check(lib().tracyhip_secondary_decomposed(self._h, C.byref(b), MEM_HOST, out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def _allelic_fraction(self, hbc, secdecomp, trim_left=50, trim_right=50, peaks_only=False):
    fr = np.zeros(2 * max(hbc.n, 1), dtype=np.float64)
    b = hbc.struct(peaks_only)
    sd = np.ascontiguousarray(secdecomp, dtype=np.uint8)
    _check(lib().tracyhip_allelic_fraction(self._h, C.byref(b), sd.ctypes.data_as(C.POINTER(C.c_uint8)), trim_left, trim_right,
                                           MEM_HOST, fr.ctypes.data_as(C.POINTER(C.c_double))))
    return fr[:2 * hbc.n].reshape(-1, 2)


Context.find_breakpoint = _find_breakpoint
Context.find_homozygous_breakpoint = _find_homozygous_breakpoint
Context.decompose_alleles = _decompose_alleles
Context.secondary_decomposed = _secondary_decomposed
Context.allelic_fraction = _allelic_fraction


class DecomposeJob(C.Structure):
    _fields_ = [("ntraces", C.c_uint32), ("profiles", SeqSet), ("bc", BaseCallsBatch), ("refs", SeqSet),
                ("ref_index", C.POINTER(C.c_uint32)), ("dprm", DecompParams), ("oriented", C.POINTER(C.c_uint8)), ("ref_profiles", SeqSet),
                ("strand_by_certificate", C.c_uint32)]


class DecomposeResult(C.Structure):
    _fields_ = [("bp", C.c_void_p), ("status", C.c_void_p), ("score_fwd", C.c_void_p), ("score_rev", C.c_void_p),
                ("forward", C.c_void_p), ("score_trim", C.c_void_p), ("dcp_indel", C.c_void_p), ("dcp_err", C.c_void_p),
                ("dcp_offset", C.POINTER(C.c_uint64)), ("dstatus", C.c_void_p), ("secdecomp", C.c_void_p),
                ("fractions", C.c_void_p), ("slice_begin", C.c_void_p * 2), ("slice_len", C.c_void_p * 2),
                ("ref_pos", C.c_void_p * 2), ("score", C.c_void_p * 3), ("ops", C.c_void_p * 3),
                ("ops_offset", C.POINTER(C.c_uint64) * 3), ("ops_len", C.c_void_p * 3)]


def _decompose_traces(self, profiles, hbc, refs, params, trim_left=50, trim_right=50, maxindel=1000, madc=5, oriented=None,
                      ref_profiles=None, exact_scores=True, peaks_only=False):
    """tracyhip_decompose_traces with host buffers; hbc: HostBaseCalls (primary/secondary rewritten in place);
    oriented: None, or rs.forward per trace when the references are already oriented (indexed-genome path)"""
    pp = profiles if isinstance(profiles, PackedSeqs) else PackedSeqs(profiles, SEQ_PROFILE)
    pr = refs if isinstance(refs, PackedSeqs) else PackedSeqs(refs, SEQ_CHAR)
    nt = pp.count
    job = DecomposeJob()
    job.ntraces = nt
    job.profiles = pp.seqset()
    job.bc = hbc.struct(peaks_only)  # (peaks_only: the peak table instead of the chromatograms, tracyhip_basecalls::peaks)
    job.refs = pr.seqset()
    job.dprm = DecompParams(trim_left, trim_right, maxindel, madc)
    job.strand_by_certificate = 0 if exact_scores else 1  # opt-in: the losing strand may carry a certified upper bound
    if oriented is not None:
        oriented = np.ascontiguousarray(oriented, dtype=np.uint8)
        job.oriented = oriented.ctypes.data_as(C.POINTER(C.c_uint8))
    if ref_profiles is not None:  # wildtype-trace reference: oriented profiles parallel to refs (= oriented primary calls)
        prp = ref_profiles if isinstance(ref_profiles, PackedSeqs) else PackedSeqs(ref_profiles, SEQ_PROFILE)
        job.ref_profiles = prp.seqset()
    cap = 2 * maxindel + 2
    doff = np.arange(max(nt, 1), dtype=np.uint64) * np.uint64(cap)
    mf = pp.length[:nt].astype(np.uint64)
    rn = pr.length[:nt].astype(np.uint64)
    res = {
        "bp": (Breakpoint * max(nt, 1))(), "status": np.zeros(max(nt, 1), np.int32),
        "score_fwd": np.zeros(max(nt, 1), np.int32), "score_rev": np.zeros(max(nt, 1), np.int32),
        "forward": np.zeros(max(nt, 1), np.uint8), "score_trim": np.zeros(max(nt, 1), np.int32),
        "dcp_indel": np.zeros(max(nt, 1) * cap, np.int32), "dcp_err": np.zeros(max(nt, 1) * cap, np.int32),
        "dstatus": (DecompStatus * max(nt, 1))(), "secdecomp": np.zeros(max(len(hbc.primary), 1), np.uint8),
        "fractions": np.zeros(2 * max(nt, 1), np.float64),
    }
    out = DecomposeResult()
    out.bp = C.addressof(res["bp"])
    out.dstatus = C.addressof(res["dstatus"])
    for k in ("status", "score_fwd", "score_rev", "forward", "score_trim", "dcp_indel", "dcp_err", "secdecomp", "fractions"):
        setattr(out, k, res[k].ctypes.data)
    out.dcp_offset = _u64p(doff)
    keep = []
    for k in range(3):
        caps = (mf + (rn if k < 2 else mf)).astype(np.uint64)
        off = np.zeros(max(nt, 1), dtype=np.uint64)
        if nt:
            off[1:nt] = np.cumsum(caps)[:-1]
        ops = np.zeros(max(int(caps.sum()), 1), np.uint8)
        olen = np.zeros(max(nt, 1), np.uint32)
        sc = np.zeros(max(nt, 1), np.int32)
        out.score[k] = sc.ctypes.data
        out.ops[k] = ops.ctypes.data
        out.ops_offset[k] = _u64p(off)
        out.ops_len[k] = olen.ctypes.data
        res["score%d" % k] = sc
        keep.append((off, ops, olen))
        if k < 2:
            for nm in ("slice_begin", "slice_len", "ref_pos"):
                a = np.zeros(max(nt, 1), np.uint32)
                getattr(out, nm)[k] = a.ctypes.data
                res["%s%d" % (nm, k)] = a
    prm = Params(params[0], params[1], params[2], params[3], 1, 0)
    if isinstance(self, Group):
        _check(lib().tracyhip_group_decompose_traces(self._g, C.byref(job), C.byref(prm), C.byref(out)))
    else:
        _check(lib().tracyhip_decompose_traces(self._h, C.byref(job), C.byref(prm), MEM_HOST, C.byref(out)))
    for k in range(3):
        off, ops, olen = keep[k]
        res["btr%d" % k] = [ops[int(off[i]):int(off[i]) + int(olen[i])].tobytes() for i in range(nt)]
    res["dcp"] = [[(int(res["dcp_indel"][i * cap + j]), int(res["dcp_err"][i * cap + j])) for j in range(res["dstatus"][i].dcp_n)]
                  for i in range(nt)]
    res["primary"] = hbc.split(hbc.primary)
    res["secondary"] = hbc.split(hbc.secondary)
    res["secdecomp_list"] = hbc.split(res["secdecomp"])
    return res


Context.decompose_traces = _decompose_traces
Context.align_banded = _align_banded
Group.align_traces = _align_traces
Group.decompose_traces = _decompose_traces
