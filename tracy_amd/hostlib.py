"""ctypes binding of the host-side C++ library (libtracy_host.so): basecalling, trace -> profile and the
seeded synthetic workloads.  These stages run on the host in the reference as well (abif.h, profile.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        p = os.path.join(_HERE, "lib", "libtracy_host.so")
        if not os.path.exists(p):
            raise ImportError("tracy_amd: %s is missing -- run `python tracy_amd/build.py`" % p)
        _LIB = C.CDLL(p)
        _LIB.tracyhost_basecall.restype = C.c_size_t
        _LIB.tracyhost_iupac.restype = C.c_char
    return _LIB


def basecall(trace, basecallpos, sigratio=0.33):
    trace = np.ascontiguousarray(trace, dtype=np.int32)
    pos = np.ascontiguousarray(basecallpos, dtype=np.int32)
    n = len(pos)
    pri = C.create_string_buffer(n + 1)
    sec = C.create_string_buffer(n + 1)
    con = C.create_string_buffer(n + 1)
    bc = np.zeros(max(n, 1), dtype=np.int32)
    k = lib().tracyhost_basecall(trace.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(trace.shape[1]),
                                 pos.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(n), C.c_float(sigratio), pri, sec,
                                 con, bc.ctypes.data_as(C.POINTER(C.c_int32)))
    return pri.raw[:k], sec.raw[:k], con.raw[:k], bc[:k].copy()


def create_profile(trace, bcpos, primary, secondary, trimleft=0, trimright=0):
    trace = np.ascontiguousarray(trace, dtype=np.int32)
    bcpos = np.ascontiguousarray(bcpos, dtype=np.int32)
    nbc = len(bcpos)
    out = np.zeros(6 * max(nbc, 1), dtype=np.float32)
    sz = lib().tracyhost_create_profile(trace.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(trace.shape[1]),
                                        bcpos.ctypes.data_as(C.POINTER(C.c_int32)), bytes(primary), bytes(secondary),
                                        C.c_size_t(nbc), int(trimleft), int(trimright),
                                        out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:6 * sz].reshape(6, sz).copy()


def synth_align(seed0, ntraces, n, mf, nthreads=0):
    """returns refs uint8 [ntraces][n], profiles float32 [ntraces][6][mf], reverse uint8 [ntraces]"""
    refs = np.zeros((ntraces, n), dtype=np.uint8)
    profs = np.zeros((ntraces, 6, mf), dtype=np.float32)
    rev = np.zeros(ntraces, dtype=np.uint8)
    lib().tracyhost_synth_align(C.c_uint64(seed0), C.c_uint32(ntraces), C.c_uint32(n), C.c_uint32(mf),
                                refs.ctypes.data_as(C.POINTER(C.c_uint8)), profs.ctypes.data_as(C.POINTER(C.c_float)),
                                rev.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint32(nthreads))
    return refs, profs, rev


def synth_decompose(seed, n, mf, maxlen=30, kind=0, frac1=0.6):
    """one synthetic heterozygous trace: returns ref (bytes), signal int32 [4][ns], basecallpos int32, indel"""
    ns = 12 * mf + 12
    ref = np.zeros(n, dtype=np.uint8)
    sig = np.zeros((4, ns), dtype=np.int32)
    pos = np.zeros(mf + 64, dtype=np.int32)
    indel = C.c_int32(0)
    lib().tracyhost_synth_decompose.restype = C.c_uint32
    npos = lib().tracyhost_synth_decompose(C.c_uint64(seed), C.c_uint32(n), C.c_uint32(mf), C.c_uint32(maxlen), int(kind),
                                           C.c_double(frac1), ref.ctypes.data_as(C.POINTER(C.c_uint8)),
                                           sig.ctypes.data_as(C.POINTER(C.c_int32)), C.c_uint32(ns),
                                           pos.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(indel))
    return ref.tobytes(), sig, pos[:npos].copy(), indel.value


def synth_decompose_batch(seed0, nt, n, mf, nthreads=0, mix=0):
    """returns dict(refs [nt][n] u8, signal [nt][4][ns] i32, bcpos [nt][mf] i32, primary/secondary [nt][mf] u8,
    profiles [nt][6][mf] f32).  mix 0: 80 % het indels, 20 % SNVs only, forward strand.  mix 1 (BASELINE configs[2] as
    SURVEY.md 8d words it): 80 % het indel + het SNVs, 10 % homozygous indel only, 10 % no variant; odd traces read the
    reverse strand of their window."""
    ns = 12 * mf + 12
    out = dict(refs=np.zeros((nt, n), np.uint8), signal=np.zeros((nt, 4, ns), np.int32), bcpos=np.zeros((nt, mf), np.int32),
               primary=np.zeros((nt, mf), np.uint8), secondary=np.zeros((nt, mf), np.uint8),
               profiles=np.zeros((nt, 6, mf), np.float32))
    lib().tracyhost_synth_decompose_batch2(C.c_uint64(seed0), C.c_uint32(nt), C.c_uint32(n), C.c_uint32(mf),
                                           out["refs"].ctypes.data_as(C.POINTER(C.c_uint8)),
                                           out["signal"].ctypes.data_as(C.POINTER(C.c_int32)),
                                           out["bcpos"].ctypes.data_as(C.POINTER(C.c_int32)),
                                           out["primary"].ctypes.data_as(C.POINTER(C.c_uint8)),
                                           out["secondary"].ctypes.data_as(C.POINTER(C.c_uint8)),
                                           out["profiles"].ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(nthreads), int(mix))
    return out


def basecall_qual(trace, basecallpos, sigratio=0.33):
    """basecall() including the estimated per-base qualities (abif.h:232-253)"""
    trace = np.ascontiguousarray(trace, dtype=np.int32)
    pos = np.ascontiguousarray(basecallpos, dtype=np.int32)
    n = len(pos)
    pri, sec, con = (C.create_string_buffer(n + 1) for _ in range(3))
    bc = np.zeros(max(n, 1), dtype=np.int32)
    q = np.zeros(max(n, 1), dtype=np.uint8)
    fn = lib().tracyhost_basecall_qual
    fn.restype = C.c_size_t
    k = fn(trace.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(trace.shape[1]), pos.ctypes.data_as(C.POINTER(C.c_int32)),
           C.c_size_t(n), C.c_float(sigratio), pri, sec, con, bc.ctypes.data_as(C.POINTER(C.c_int32)),
           q.ctypes.data_as(C.POINTER(C.c_uint8)))
    return pri.raw[:k], sec.raw[:k], con.raw[:k], bc[:k].copy(), q[:k].copy()


def _read_trace_with(l, prefix, path, with_format):
    rd = getattr(l, prefix + "trace_read")
    rd.restype = C.c_void_p
    fmt = C.c_int32(-1)
    h = rd(os.fsencode(path), C.byref(fmt)) if with_format else rd(os.fsencode(path))
    if not h:
        return None
    h = C.c_void_p(h)
    try:
        ns, nc = C.c_uint64(0), C.c_uint64(0)
        getattr(l, prefix + "trace_dims")(h, C.byref(ns), C.byref(nc))
        sig = np.zeros((4, max(ns.value, 1)), dtype=np.int32)
        pos = np.zeros(max(nc.value, 1), dtype=np.int32)
        b1, b2 = C.create_string_buffer(nc.value + 1), C.create_string_buffer(nc.value + 1)
        q = np.zeros(max(nc.value, 1), dtype=np.uint8)
        getattr(l, prefix + "trace_get")(h, sig.ctypes.data_as(C.POINTER(C.c_int32)), pos.ctypes.data_as(C.POINTER(C.c_int32)), b1, b2,
                                         q.ctypes.data_as(C.POINTER(C.c_uint8)))
        return dict(format=fmt.value, signal=sig[:, :ns.value].copy(), basecallpos=pos[:nc.value].copy(), basecalls1=b1.raw[:nc.value],
                    basecalls2=b2.raw[:nc.value], qual=q[:nc.value].copy())
    finally:
        getattr(l, prefix + "trace_free")(h)


def read_trace(path):
    """ABIF (readab, abif.h:286-405) or SCF (readscf, scf.h:38-102) chromatogram -> dict, None when unreadable"""
    return _read_trace_with(lib(), "tracyhost_", path, True)


def write_abif(path, signal, peaks, primary, qual, secondary=b"", order=b"GATC"):
    """the build's own ABIF writer: signal int32 [4][ns] in A,C,G,T order"""
    signal = np.ascontiguousarray(signal, dtype=np.int32)
    peaks = np.ascontiguousarray(peaks, dtype=np.int32)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    rc = lib().tracyhost_writeab(os.fsencode(path), signal.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(signal.shape[1]),
                                 peaks.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(len(peaks)), bytes(primary),
                                 C.c_size_t(len(primary)), qual.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(len(qual)),
                                 bytes(secondary) if secondary else None, C.c_size_t(len(secondary)), bytes(order))
    if rc != 0:
        raise IOError("tracy_amd: cannot write %s" % path)


def trace_txt(outfile, trace, basecallpos, sigratio=0.33, left_trim=0, right_trim=0):
    """basecall + traceTxtOut (abif.h:513-533)"""
    trace = np.ascontiguousarray(trace, dtype=np.int32)
    pos = np.ascontiguousarray(basecallpos, dtype=np.int32)
    return lib().tracyhost_trace_txt(os.fsencode(outfile), trace.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(trace.shape[1]),
                                     pos.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(len(pos)), C.c_float(sigratio),
                                     C.c_uint32(left_trim), C.c_uint32(right_trim))


def align_outputs(prefix, trace_stem, trace, basecallpos, sigratio, row0, row1, chrom, refslice, pos, forward, score, linelimit=60):
    """writes <prefix>.align.fa / .txt / .json exactly as `tracy align` does (sage.h:313-345)"""
    trace = np.ascontiguousarray(trace, dtype=np.int32)
    bp = np.ascontiguousarray(basecallpos, dtype=np.int32)
    return lib().tracyhost_align_outputs(os.fsencode(prefix), trace_stem.encode(), trace.ctypes.data_as(C.POINTER(C.c_int32)),
                                         C.c_size_t(trace.shape[1]), bp.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(len(bp)),
                                         C.c_float(sigratio), bytes(row0), bytes(row1), C.c_size_t(len(row0)), chrom.encode(),
                                         bytes(refslice), C.c_size_t(len(refslice)), C.c_uint32(pos), int(bool(forward)), int(score),
                                         C.c_uint32(linelimit))


def trim_trace(trace, basecallpos, sigratio, stringency):
    trace = np.ascontiguousarray(trace, dtype=np.int32)
    bp = np.ascontiguousarray(basecallpos, dtype=np.int32)
    l, r = C.c_uint32(0), C.c_uint32(0)
    lib().tracyhost_trim_trace(trace.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(trace.shape[1]), bp.ctypes.data_as(C.POINTER(C.c_int32)),
                               C.c_size_t(len(bp)), C.c_float(sigratio), C.c_float(stringency), C.byref(l), C.byref(r))
    return l.value, r.value


def load_fasta(path):
    """loadSingleFasta (fasta.h:54-95) -> (name, seq) or None"""
    name = C.create_string_buffer(4096)
    cap = os.path.getsize(path) + 16
    seq = C.create_string_buffer(cap)
    fn = lib().tracyhost_load_fasta
    fn.restype = C.c_int64
    n = fn(os.fsencode(path), name, C.c_size_t(4096), seq, C.c_size_t(cap))
    if n < 0:
        return None
    return name.value.decode("latin1"), seq.raw[:n].decode()


class DecomposeReport(C.Structure):
    _fields_ = [("prefix", C.c_char_p), ("genome_name", C.c_char_p), ("input_name", C.c_char_p), ("trace", C.POINTER(C.c_int32)),
                ("nsamples", C.c_uint64), ("basecallpos", C.POINTER(C.c_int32)), ("npos", C.c_uint64), ("pratio", C.c_float),
                ("trim_left", C.c_uint32), ("trim_right", C.c_uint32), ("qual_cut", C.c_uint32), ("linelimit", C.c_uint32),
                ("primary", C.c_char_p), ("secondary", C.c_char_p), ("secdecomp", C.c_char_p), ("ncalls", C.c_uint64),
                ("rows", (C.c_char_p * 2) * 3), ("cols", C.c_uint64 * 3), ("var_rows", (C.c_char_p * 2) * 2), ("var_cols", C.c_uint64 * 2),
                ("call_variants", C.c_int32), ("chr", C.c_char_p), ("forward", C.c_int32), ("pos", C.c_uint32 * 2),
                ("slice_len", C.c_uint64 * 2), ("ref_len", C.c_uint64), ("score", C.c_int32 * 3), ("indelshift", C.c_int32),
                ("breakpoint", C.c_uint32), ("a1", C.c_double), ("a2", C.c_double), ("dcp_indel", C.POINTER(C.c_int32)),
                ("dcp_err", C.POINTER(C.c_int32)), ("dcp_n", C.c_uint64)]


def decompose_outputs(prefix, genome_name, input_name, trace, basecallpos, pratio, trims, qual_cut, linelimit, primary, secondary, secdecomp,
                      rows, var_rows, chrom, forward, pos, slice_len, ref_len, scores, indelshift, breakpoint, a1a2, dcp):
    """writes the files of `tracy decompose` (indigo.h:340-442); rows: 3 x (row0, row1); var_rows: None or 2 x (row0, row1)"""
    trace = np.ascontiguousarray(trace, dtype=np.int32)
    bp = np.ascontiguousarray(basecallpos, dtype=np.int32)
    di = np.array([a for a, _ in dcp], dtype=np.int32)
    de = np.array([b for _, b in dcp], dtype=np.int32)
    r = DecomposeReport()
    r.prefix, r.genome_name, r.input_name, r.chr = os.fsencode(prefix), genome_name.encode(), input_name.encode(), chrom.encode()
    r.trace, r.nsamples = trace.ctypes.data_as(C.POINTER(C.c_int32)), trace.shape[1]
    r.basecallpos, r.npos, r.pratio = bp.ctypes.data_as(C.POINTER(C.c_int32)), len(bp), pratio
    r.trim_left, r.trim_right, r.qual_cut, r.linelimit = trims[0], trims[1], qual_cut, linelimit
    r.primary, r.secondary, r.secdecomp, r.ncalls = bytes(primary), bytes(secondary), bytes(secdecomp), len(primary)
    for k in range(3):
        r.rows[k][0], r.rows[k][1], r.cols[k] = bytes(rows[k][0]), bytes(rows[k][1]), len(rows[k][0])
    r.call_variants = 1 if var_rows else 0
    if var_rows:
        for k in range(2):
            r.var_rows[k][0], r.var_rows[k][1], r.var_cols[k] = bytes(var_rows[k][0]), bytes(var_rows[k][1]), len(var_rows[k][0])
    r.forward, r.ref_len, r.indelshift, r.breakpoint = int(bool(forward)), ref_len, int(bool(indelshift)), breakpoint
    for k in range(2):
        r.pos[k], r.slice_len[k] = pos[k], slice_len[k]
    for k in range(3):
        r.score[k] = scores[k]
    r.a1, r.a2 = a1a2
    r.dcp_indel, r.dcp_err, r.dcp_n = di.ctypes.data_as(C.POINTER(C.c_int32)), de.ctypes.data_as(C.POINTER(C.c_int32)), len(dcp)
    return lib().tracyhost_decompose_outputs(C.byref(r))


class Genome:
    """indexed genome for k-mer seeding (tracy_amd/host/seed.hpp): plain or gzip-compressed multi-FASTA (table built in memory), or an
    index file written by save() / `tracy_amd_cli index` (mapped read-only; ranks of one node share the page cache's copy)"""

    def __init__(self, path, kmer=15, nthreads=0):
        self._h = None
        fn = lib().tracyhost_genome_open
        fn.restype = C.c_void_p
        h = fn(os.fsencode(path), C.c_uint32(kmer), C.c_uint32(nthreads))
        if not h:
            raise IOError("tracy_amd: cannot read genome %s" % path)
        self._h = C.c_void_p(h)
        self.kmer = kmer

    def close(self):
        if self._h:
            lib().tracyhost_genome_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def save(self, path):
        """write the index (text, contigs, sorted k-mer table) to one file that Genome(path) maps read-only: `tracy index`, index.h:79-124"""
        if lib().tracyhost_genome_save(self._h, os.fsencode(path)) != 0:
            raise IOError("tracy_amd: cannot write index %s" % path)

    def count(self, pattern):
        fn = lib().tracyhost_genome_count
        fn.restype = C.c_uint64
        return int(fn(self._h, bytes(pattern), C.c_size_t(len(pattern))))

    @staticmethod
    def pack_consensus(consensus):
        """the batch as a C caller holds it: one byte block, offsets, lengths (seed_packed takes these; packing a list of Python
        strings is interpreter work, not part of the seeding stage)"""
        n = len(consensus)
        lens = np.array([len(c) for c in consensus], dtype=np.uint32)
        offs = np.zeros(max(n, 1), dtype=np.uint64)
        if n:
            offs[1:n] = np.cumsum(lens.astype(np.uint64))[:-1]
        return dict(n=n, blob=b"".join(bytes(c) for c in consensus) + b"\0", offs=offs, lens=lens)

    def seed_packed(self, packed, trim_left=50, trim_right=50, min_support=3, maxindel=1000, nthreads=0, out=None):
        """getReferenceSlice for a packed batch (pack_consensus); `out`: the dict a previous call returned, reused as the result
        buffers of this one (same batch shape) -- what a caller that seeds block after block does.  Returns the raw form of seed()."""
        n = packed["n"]
        cap = int(packed["lens"].max() if n else 0) + 2 * maxindel + 2
        if out is None or out["slices_2d"].shape != (max(n, 1), cap):
            out = dict(status=np.zeros(max(n, 1), np.int32), forward=np.zeros(max(n, 1), np.uint8), kmersupport=np.zeros(max(n, 1), np.uint32),
                       pos=np.zeros(max(n, 1), np.uint32), contig=np.zeros(max(n, 1), np.uint32), slice_len=np.zeros(max(n, 1), np.uint32),
                       slices_2d=np.zeros((max(n, 1), cap), dtype=np.uint8))
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        lib().tracyhost_seed_batch(self._h, C.c_uint32(n), packed["blob"], p(packed["offs"], C.c_uint64), p(packed["lens"], C.c_uint32), C.c_uint32(trim_left),
                                   C.c_uint32(trim_right), C.c_uint32(self.kmer), C.c_uint32(min_support), C.c_uint32(maxindel),
                                   C.c_uint32(nthreads), p(out["status"], C.c_int32), p(out["forward"], C.c_uint8),
                                   p(out["kmersupport"], C.c_uint32), p(out["pos"], C.c_uint32), p(out["contig"], C.c_uint32),
                                   out["slices_2d"].ctypes.data_as(C.c_char_p), C.c_uint64(cap), p(out["slice_len"], C.c_uint32))
        return out

    def seed(self, consensus, trim_left=50, trim_right=50, min_support=3, maxindel=1000, nthreads=0, raw=False):
        """getReferenceSlice (fmindex.h:236-326) for a list of consensus strings -> dict of arrays + oriented windows
        (`slices`: list of bytes; raw=True: `slices_2d` uint8 [n][cap] + `slice_len` instead, no per-trace copies)"""
        n = len(consensus)
        lens = np.array([len(c) for c in consensus], dtype=np.uint32)
        offs = np.zeros(max(n, 1), dtype=np.uint64)
        if n:
            offs[1:n] = np.cumsum(lens.astype(np.uint64))[:-1]
        blob = b"".join(bytes(c) for c in consensus) + b"\0"
        cap = int(lens.max() if n else 0) + 2 * maxindel + 2
        out = dict(status=np.zeros(max(n, 1), np.int32), forward=np.zeros(max(n, 1), np.uint8), kmersupport=np.zeros(max(n, 1), np.uint32),
                   pos=np.zeros(max(n, 1), np.uint32), contig=np.zeros(max(n, 1), np.uint32), slice_len=np.zeros(max(n, 1), np.uint32))
        slices = np.zeros((max(n, 1), cap), dtype=np.uint8)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        lib().tracyhost_seed_batch(self._h, C.c_uint32(n), blob, p(offs, C.c_uint64), p(lens, C.c_uint32), C.c_uint32(trim_left),
                                   C.c_uint32(trim_right), C.c_uint32(self.kmer), C.c_uint32(min_support), C.c_uint32(maxindel),
                                   C.c_uint32(nthreads), p(out["status"], C.c_int32), p(out["forward"], C.c_uint8),
                                   p(out["kmersupport"], C.c_uint32), p(out["pos"], C.c_uint32), p(out["contig"], C.c_uint32),
                                   slices.ctypes.data_as(C.c_char_p), C.c_uint64(cap), p(out["slice_len"], C.c_uint32))
        out = {k: v[:n] for k, v in out.items()}
        if raw:
            out["slices_2d"] = slices
        else:
            out["slices"] = [slices[i, :out["slice_len"][i]].tobytes() for i in range(n)]
        return out
