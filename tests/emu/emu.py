"""ctypes driver of the host wave emulator (tests only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = None

MODE_CHAR, MODE_QP, MODE_PROF, MODE_CQ = 0, 1, 2, 3


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libemu_wave.so")
        srcs = [os.path.join(_HERE, "emu_wave.cpp"), os.path.join(_ROOT, "tracy_amd/csrc/dp_kernels.h"),
                os.path.join(_ROOT, "tracy_amd/csrc/dp_lane.h"), os.path.join(_ROOT, "tracy_amd/csrc/band16.h"),
                os.path.join(_ROOT, "tracy_amd/csrc/front.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-std=c++20", "-O1", "-fPIC", "-shared", "-pthread", "-ffp-contract=off",
                                   "-o", so, srcs[0]], stderr=subprocess.DEVNULL)
        _LIB = C.CDLL(so)
    return _LIB


def run(a1, a2, score, hfree, vfree, mode, K, trace=True, needle=False, revcomp=False, a1_view=None, narrow=False, screen=False, band=None):
    """a1/a2: bytes or float32 [6][len] arrays.  a1_view=(offset, m): use columns [offset, offset+m) of a1."""
    def prep(x):
        if isinstance(x, (bytes, bytearray)):
            buf = np.frombuffer(bytes(x) + b"\0", dtype=np.uint8).copy()
            return buf, len(x), len(x)
        x = np.ascontiguousarray(x, dtype=np.float32)
        return x, x.shape[1], x.shape[1]
    b1, m, s1 = prep(a1)
    if mode == MODE_QP:  # the API layer encodes reference characters into profile-row codes
        lut = np.full(256, 6, dtype=np.uint8)
        for chars, code in ((b"Aa", 0), (b"Cc", 1), (b"Gg", 2), (b"Tt", 3), (b"Nn", 4), (b"-", 5)):
            for ch in chars:
                lut[ch] = code
        a2 = lut[np.frombuffer(bytes(a2), dtype=np.uint8)].tobytes()
    b2, n, s2 = prep(a2)
    p1 = b1.ctypes.data
    if a1_view is not None:
        off, m = a1_view
        p1 += off * b1.itemsize
    sc = C.c_int32(0)
    ops = C.create_string_buffer(m + n + 2)
    ol = C.c_uint32(0)
    err = C.c_int32(0)
    rc = lib().emu_dp(int(needle), mode, K, int(trace), C.c_void_p(p1), m, s1, C.c_void_p(b2.ctypes.data), n, s2,
                      (1 if revcomp else 0) | (0x100 if narrow else 0) | (0x200 if screen else 0) | ((0x400 | (int(band) << 16)) if band is not None else 0), *[int(x) for x in score], int(hfree), int(vfree), C.byref(sc), ops,
                      C.byref(ol), C.byref(err))
    assert rc == 0
    return sc.value, (ops.raw[:ol.value] if trace else None), err.value


def run_band(a1, a2, score, hfree, vfree, mode, K, B=32, narrow=True, revcomp=False, a1_view=None):
    """checkpointed score kernel + band traceback of one pair; returns (score, btr, err)"""
    if isinstance(a1, (bytes, bytearray)):
        b1 = np.frombuffer(bytes(a1) + b"\0", dtype=np.uint8).copy()
        m, s1 = len(a1), len(a1)
    else:
        b1 = np.ascontiguousarray(a1, dtype=np.float32)
        m, s1 = b1.shape[1], b1.shape[1]
    if mode == MODE_QP:
        lut = np.full(256, 6, dtype=np.uint8)
        for chars, code in ((b"Aa", 0), (b"Cc", 1), (b"Gg", 2), (b"Tt", 3), (b"Nn", 4), (b"-", 5)):
            for ch in chars:
                lut[ch] = code
        a2 = lut[np.frombuffer(bytes(a2), dtype=np.uint8)].tobytes()
    b2 = np.frombuffer(bytes(a2) + b"\0", dtype=np.uint8).copy()
    n = len(a2)
    p1 = b1.ctypes.data
    if a1_view is not None:
        off, m = a1_view
        p1 += off * b1.itemsize
    sc = C.c_int32(0)
    ops = C.create_string_buffer(m + n + 2)
    ol = C.c_uint32(0)
    err = C.c_int32(0)
    rc = lib().emu_band(mode, K, int(narrow), B, C.c_void_p(p1), m, s1, C.c_void_p(b2.ctypes.data), n, 1 if revcomp else 0,
                        *[int(x) for x in score], int(hfree), int(vfree), C.byref(sc), ops, C.byref(ol), C.byref(err))
    assert rc == 0
    return sc.value, ops.raw[:ol.value], err.value


def run_prefix(profiles, refs, score, K, revcomp=None):
    """prefix-bound kernel body on one emulated wave: up to 8 pairs (profile float32 [6][m], reference bytes).
    Returns the list of max_j max(H, F)(R, j), R = 8*K, and the error flags."""
    npairs = len(profiles)
    assert npairs <= 8
    lut = np.full(256, 6, dtype=np.uint8)
    for chars, code in ((b"Aa", 0), (b"Cc", 1), (b"Gg", 2), (b"Tt", 3), (b"Nn", 4), (b"-", 5)):
        for ch in chars:
            lut[ch] = code
    a1 = np.concatenate([np.ascontiguousarray(p, dtype=np.float32).ravel() for p in profiles])
    a1_off = np.zeros(npairs, np.uint64)
    m = np.array([p.shape[1] for p in profiles], np.uint32)
    if npairs > 1:
        a1_off[1:] = np.cumsum(6 * m.astype(np.uint64))[:-1]
    a2 = np.concatenate([lut[np.frombuffer(bytes(r), dtype=np.uint8)] for r in refs] + [np.zeros(1, np.uint8)])
    n = np.array([len(r) for r in refs], np.uint32)
    a2_off = np.zeros(npairs, np.uint64)
    if npairs > 1:
        a2_off[1:] = np.cumsum(n.astype(np.uint64))[:-1]
    flags = np.array([1 if (revcomp and revcomp[i]) else 0 for i in range(npairs)], np.uint32)
    out = np.zeros(npairs, np.int32)
    err = C.c_int32(0)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = lib().emu_prefix(K, npairs, p(a1, C.c_float), p(a1_off, C.c_uint64), p(m, C.c_uint32), p(a2, C.c_uint8), p(a2_off, C.c_uint64),
                          p(n, C.c_uint32), p(flags, C.c_uint32), *[int(x) for x in score], p(out, C.c_int32), C.byref(err))
    assert rc == 0
    return out.tolist(), err.value


def run_origin(a1, a2, score, K, revcomp=False, table=False):
    """origin-tracking sweep body on one emulated wave (string x string, semiglobal): (score, leading 'h' columns,
    last column that is not a trailing 'h')"""
    m, n = len(a1), len(a2)
    b1 = np.frombuffer(bytes(a1) + b"\0", dtype=np.uint8).copy()
    b2 = np.frombuffer(bytes(a2) + b"\0", dtype=np.uint8).copy()
    sc = C.c_int32(0)
    ends = np.zeros(2, np.uint32)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = lib().emu_origin(K, p(b1, C.c_uint8), C.c_uint32(m), p(b2, C.c_uint8), C.c_uint32(n), C.c_uint32((1 if revcomp else 0) | (0x200 if table else 0)),
                          *[int(x) for x in score], C.byref(sc), p(ends, C.c_uint32))
    assert rc == 0
    return sc.value, int(ends[0]), int(ends[1])


def screen_check(a, b, match, mismatch, nt):
    """a, b: float32 [n][5] profile columns.  Returns (strips the screened score could not prove, proven strips whose int
    differs from the exact float chain) -- SubProf::screen against SubProf::prepare (dp_kernels.h)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape and a.shape[1] == 5
    counts = (C.c_uint64 * 2)()
    rc = lib().emu_screen_check(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_uint64(a.shape[0]), int(match), int(mismatch),
                                int(nt), counts)
    assert rc == 0
    return int(counts[0]), int(counts[1])


def run_origin_qp(prof, a2, score, K, revcomp=False):
    """origin-tracking sweep with profile rows (the preliminary alignment of `tracy align`): prof float32 [6][m], a2 reference
    characters (encoded as the library does).  Returns (score, leading 'h' columns, last column that is not a trailing 'h')"""
    prof = np.ascontiguousarray(prof, dtype=np.float32)
    m = prof.shape[1]
    lut = np.full(256, 6, dtype=np.uint8)
    for chars, code in ((b"Aa", 0), (b"Cc", 1), (b"Gg", 2), (b"Tt", 3), (b"Nn", 4), (b"-", 5)):
        for ch in chars:
            lut[ch] = code
    b2 = np.frombuffer(lut[np.frombuffer(bytes(a2), dtype=np.uint8)].tobytes() + b"\0", dtype=np.uint8).copy()
    n = len(a2)
    sc = C.c_int32(0)
    ends = np.zeros(2, np.uint32)
    rc = lib().emu_origin_qp(K, C.c_void_p(prof.ctypes.data), C.c_uint32(m), C.c_uint32(m), b2.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint32(n),
                             C.c_uint32(1 if revcomp else 0), *[int(x) for x in score], C.byref(sc), ends.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == 0
    return sc.value, int(ends[0]), int(ends[1])


def run_band16(pairs, score, hfree, K, kind=0, strings=True):
    """pairs: up to four of (a1, a2, dmin, dmax, revcomp); a1 bytes (strings) or float32 [6][m] profiles, a2 reference characters.
    Returns per pair (score, btr or None, ends or None) and the error word."""
    npairs = len(pairs)
    a1_off, a1_stride, ms, a2_off, ns, flags, dmins, dmaxs = [], [], [], [], [], [], [], []
    if strings:
        blob1 = b"".join(bytes(p[0]) for p in pairs) + b"\0"
        o = 0
        for p in pairs:
            a1_off.append(o); a1_stride.append(len(p[0])); ms.append(len(p[0])); o += len(p[0])
        buf1 = np.frombuffer(blob1, dtype=np.uint8).copy()
    else:
        arrs = [np.ascontiguousarray(p[0], dtype=np.float32) for p in pairs]
        o = 0
        for x in arrs:
            a1_off.append(o); a1_stride.append(x.shape[1]); ms.append(x.shape[1]); o += x.size
        buf1 = np.concatenate([x.ravel() for x in arrs] + [np.zeros(1, np.float32)])
    blob2 = b"".join(bytes(p[1]) for p in pairs) + b"\0"
    o = 0
    for p in pairs:
        a2_off.append(o); ns.append(len(p[1])); o += len(p[1])
        dmins.append(int(p[2])); dmaxs.append(int(p[3])); flags.append(1 if (len(p) > 4 and p[4]) else 0)
    buf2 = np.frombuffer(blob2, dtype=np.uint8).copy()
    cap = max(m + n for m, n in zip(ms, ns)) + 2
    u64 = lambda v: np.asarray(v, dtype=np.uint64)
    u32 = lambda v: np.asarray(v, dtype=np.uint32)
    i32 = lambda v: np.asarray(v, dtype=np.int32)
    A = dict(a1_off=u64(a1_off), a1_stride=u32(a1_stride), m=u32(ms), a2_off=u64(a2_off), n=u32(ns), flags=u32(flags), dmin=i32(dmins), dmax=i32(dmaxs))
    scores = np.zeros(16, np.int32)  # (K = 44, the quad form: up to sixteen pairs)
    ends = np.zeros(32, np.uint32)
    ops = np.zeros(16 * cap, np.uint8)
    ops_len = np.zeros(16, np.uint32)
    err = C.c_int32(0)
    ptr = lambda x: C.c_void_p(x.ctypes.data)
    rc = lib().emu_band16(int(K), int(kind), 1 if strings else 0, npairs, ptr(buf1), ptr(A["a1_off"]), ptr(A["a1_stride"]), ptr(A["m"]), ptr(buf2),
                          ptr(A["a2_off"]), ptr(A["n"]), ptr(A["flags"]), ptr(A["dmin"]), ptr(A["dmax"]), *[int(x) for x in score], int(hfree),
                          ptr(scores), ptr(ends), ptr(ops), C.c_uint64(cap), ptr(ops_len), C.byref(err))
    assert rc == 0
    out = []
    for i in range(npairs):
        btr = ops[i * cap:i * cap + int(ops_len[i])].tobytes() if kind == 0 else None
        out.append((int(scores[i]), btr, (int(ends[2 * i]), int(ends[2 * i + 1])) if kind == 1 else None))
    return out, err.value


def run_front(profiles, refs, score, Kp, Kb, halfw, revcomp=None, want_rows=False, GLp=8, second_bound=True, cont16=False, quad=False):
    """the pruned orientation sweep (front.h) of up to four traces: prefix rows kept, band placed, band swept below the kept row,
    certificate.  Returns per pair a dict(vmax, cstar, shift, ok, score, c_e[, row]) and the error word."""
    npairs = len(profiles)
    assert 1 <= npairs <= 4
    a1 = np.concatenate([np.ascontiguousarray(p, dtype=np.float32).ravel() for p in profiles] + [np.zeros(1, np.float32)])
    m = np.array([p.shape[1] for p in profiles], np.uint32)
    a1_off = np.zeros(npairs, np.uint64)
    if npairs > 1:
        a1_off[1:] = np.cumsum(6 * m.astype(np.uint64))[:-1]
    a2 = np.frombuffer(b"".join(bytes(r) for r in refs) + b"\0", dtype=np.uint8).copy()
    n = np.array([len(r) for r in refs], np.uint32)
    a2_off = np.zeros(npairs, np.uint64)
    if npairs > 1:
        a2_off[1:] = np.cumsum(n.astype(np.uint64))[:-1]
    flags = np.array([1 if (revcomp and revcomp[i]) else 0 for i in range(npairs)], np.uint32)
    out = np.zeros(6 * npairs, np.int32)
    cap = int(n.max()) + 1
    rows = np.zeros(npairs * cap, np.uint32)
    err = C.c_int32(0)
    ptr = lambda x: C.c_void_p(x.ctypes.data)
    rc = lib().emu_front(int(Kp), int(GLp), int(Kb), npairs, ptr(a1), ptr(a1_off), ptr(m), ptr(a2), ptr(a2_off), ptr(n), ptr(flags), int(halfw),
                         *[int(x) for x in score], ptr(out), ptr(rows) if want_rows else None, C.c_uint64(cap), C.byref(err), C.c_int32((1 if second_bound else 0) | (2 if cont16 else 0) | (4 if quad else 0)))  # (bit 1: the band on the 16-bit cells; bit 2: in the quad form)
    assert rc == 0, rc
    res = []
    for i in range(npairs):
        o = out[6 * i:6 * i + 6]
        r = dict(vmax=int(o[0]), cstar=int(o[1]), shift=int(o[2]), ok=int(o[3]), score=int(o[4]), c_e=int(o[5]))
        if want_rows:
            r["row"] = rows[i * cap:i * cap + int(n[i]) + 1].copy()
        res.append(r)
    return res, err.value


def table_rows(profile, score):
    """int substitution scores of a profile's rows against the codes A C G T N other: array [m][6]"""
    x = np.ascontiguousarray(profile, dtype=np.float32)
    m = x.shape[1]
    out = np.zeros((m, 6), np.int32)
    rc = lib().emu_table_rows(C.c_void_p(x.ctypes.data), C.c_uint64(0), C.c_uint32(m), C.c_uint32(m), int(score[0]), int(score[1]), C.c_void_p(out.ctypes.data))
    assert rc == 0
    return out
