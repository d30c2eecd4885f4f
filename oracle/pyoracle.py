"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (tracy_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


class Score(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("go", C.c_int32), ("ge", C.c_int32)]


class Breakpoint(C.Structure):
    _fields_ = [("indelshift", C.c_int32), ("traceleft", C.c_int32), ("breakpoint", C.c_uint32),
                ("bestDiff", C.c_float)]


class DecompCfg(C.Structure):
    _fields_ = [("trimLeft", C.c_int32), ("trimRight", C.c_int32), ("maxindel", C.c_int32), ("madc", C.c_int32)]


class DecompStatus(C.Structure):
    _fields_ = [("kind", C.c_int32), ("bestIns", C.c_int32), ("bestDel", C.c_int32), ("bestFR", C.c_int32)]


class TrimResult(C.Structure):
    _fields_ = [("ri", C.c_uint32), ("risize", C.c_uint32), ("pos_add", C.c_uint32), ("warn", C.c_int32)]


def build(force=False):
    so = os.path.join(_HERE, "libtracy_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("tracy_oracle.c", "tracy_oracle_decompose.c", "tracy_oracle_abif.c",
                                             "tracy_oracle_chain.c", "tracy_oracle.h", "tracy_oracle_decompose.h")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "libtracy_oracle.so"], stdout=subprocess.DEVNULL)
    # oracle/_ref is (re)built only where the reference sources exist (this container)
    if os.path.exists("/root/reference/src/abif.h"):
        ref = os.path.join(_HERE, "_ref", "libref_abif.so")
        if force or not os.path.exists(ref) or os.path.getmtime(os.path.join(_HERE, "ref_abif.cpp")) > os.path.getmtime(ref):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_profile_cons_char.restype = C.c_char
        _LIB.orc_iupac2.restype = C.c_char
        _LIB.orc_basecall.restype = C.c_size_t
    return _LIB


def ref_lib():
    """The reference's own abif.h compiled into oracle/_ref (None when it was never built)."""
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libref_abif.so")
        if not os.path.exists(p):
            return None
        _REF = C.CDLL(p)
        _REF.ref_basecall.restype = C.c_size_t
        _REF.ref_iupac2.restype = C.c_char
        _REF.ref_trimmed_seq.restype = C.c_size_t
    return _REF


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode()


def _fp(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def _sc(score):
    return Score(*[int(x) for x in score])


# ---- DP -------------------------------------------------------------------------------------
def gotoh_score_str(s1, s2, hfree, vfree, score):
    s1, s2 = _b(s1), _b(s2)
    return lib().orc_gotoh_score_str(s1, C.c_size_t(len(s1)), s2, C.c_size_t(len(s2)), int(hfree), int(vfree),
                                     C.byref(_sc(score)))


def _run_trace(fn, a1, l1, a2, l2, hfree, vfree, score):
    btr = C.create_string_buffer(l1 + l2 + 1)
    n = C.c_size_t(0)
    sc = fn(a1, C.c_size_t(l1), a2, C.c_size_t(l2), int(hfree), int(vfree), C.byref(_sc(score)), btr, C.byref(n))
    return sc, btr.raw[:n.value]


def gotoh_str(s1, s2, hfree, vfree, score):
    """returns (score, btr) -- btr in the reference's push order (back to front)"""
    s1, s2 = _b(s1), _b(s2)
    return _run_trace(lib().orc_gotoh_str, s1, len(s1), s2, len(s2), hfree, vfree, score)


def gotoh_score_prof(p1, p2, hfree, vfree, score):
    p1, q1 = _fp(p1)
    p2, q2 = _fp(p2)
    return lib().orc_gotoh_score_prof(q1, C.c_size_t(p1.shape[1]), q2, C.c_size_t(p2.shape[1]), int(hfree),
                                      int(vfree), C.byref(_sc(score)))


def gotoh_prof(p1, p2, hfree, vfree, score):
    p1, q1 = _fp(p1)
    p2, q2 = _fp(p2)
    return _run_trace(lib().orc_gotoh_prof, q1, p1.shape[1], q2, p2.shape[1], hfree, vfree, score)


def needle_score_str(s1, s2, hfree, vfree, score):
    s1, s2 = _b(s1), _b(s2)
    return lib().orc_needle_score_str(s1, C.c_size_t(len(s1)), s2, C.c_size_t(len(s2)), int(hfree), int(vfree),
                                      C.byref(_sc(score)))


def needle_str(s1, s2, hfree, vfree, score):
    s1, s2 = _b(s1), _b(s2)
    return _run_trace(lib().orc_needle_str, s1, len(s1), s2, len(s2), hfree, vfree, score)


def needle_score_prof(p1, p2, hfree, vfree, score):
    p1, q1 = _fp(p1)
    p2, q2 = _fp(p2)
    return lib().orc_needle_score_prof(q1, C.c_size_t(p1.shape[1]), q2, C.c_size_t(p2.shape[1]), int(hfree),
                                       int(vfree), C.byref(_sc(score)))


def needle_prof(p1, p2, hfree, vfree, score):
    p1, q1 = _fp(p1)
    p2, q2 = _fp(p2)
    return _run_trace(lib().orc_needle_prof, q1, p1.shape[1], q2, p2.shape[1], hfree, vfree, score)


def create_alignment_str(btr, s1, s2):
    s1, s2 = _b(s1), _b(s2)
    r0 = C.create_string_buffer(len(btr) + 1)
    r1 = C.create_string_buffer(len(btr) + 1)
    lib().orc_create_alignment_str(btr, C.c_size_t(len(btr)), s1, s2, r0, r1)
    return r0.raw[:len(btr)], r1.raw[:len(btr)]


def create_alignment_prof(btr, p1, p2):
    p1, q1 = _fp(p1)
    p2, q2 = _fp(p2)
    r0 = C.create_string_buffer(len(btr) + 1)
    r1 = C.create_string_buffer(len(btr) + 1)
    lib().orc_create_alignment_prof(btr, C.c_size_t(len(btr)), q1, C.c_size_t(p1.shape[1]), q2,
                                    C.c_size_t(p2.shape[1]), r0, r1)
    return r0.raw[:len(btr)], r1.raw[:len(btr)]


# ---- profiles --------------------------------------------------------------------------------
def create_profile_str(s):
    s = _b(s)
    p = np.zeros((6, len(s)), dtype=np.float32)
    lib().orc_create_profile_str(s, C.c_size_t(len(s)), p.ctypes.data_as(C.POINTER(C.c_float)))
    return p


def create_profile_trace(trace, bcpos, primary, secondary, trimleft=0, trimright=0):
    trace, tp = _ip(trace)
    bcpos, bp = _ip(bcpos)
    primary, secondary = _b(primary), _b(secondary)
    nbc = len(bcpos)
    p = np.zeros(6 * max(nbc, 1), dtype=np.float32)
    sz = lib().orc_create_profile_trace(tp, C.c_size_t(trace.shape[1]), bp, primary, secondary, C.c_size_t(nbc),
                                        int(trimleft), int(trimright), p.ctypes.data_as(C.POINTER(C.c_float)))
    return p[:6 * sz].reshape(6, sz).copy()


def revcomp_profile(p):
    p, q = _fp(p)
    out = np.zeros_like(p)
    lib().orc_revcomp_profile(q, C.c_size_t(p.shape[1]), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def trim_reference_slice(row0, row1, trim_left, trim_right, refslice_size, forward=True):
    out = TrimResult()
    lib().orc_trim_reference_slice(_b(row0), _b(row1), C.c_size_t(len(row0)), C.c_uint32(trim_left),
                                   C.c_uint32(trim_right), C.c_size_t(refslice_size), int(forward), C.byref(out))
    return out.ri, out.risize, out.pos_add, out.warn


# ---- decompose -------------------------------------------------------------------------------
def find_breakpoint(ptrace):
    p, q = _fp(ptrace)
    bp = Breakpoint()
    lib().orc_find_breakpoint(q, C.c_size_t(p.shape[1]), C.byref(bp))
    return bp


def find_homozygous_breakpoint(row0, row1, bp=None):
    bp = bp or Breakpoint()
    rc = lib().orc_find_homozygous_breakpoint(_b(row0), _b(row1), C.c_size_t(len(row0)), C.byref(bp))
    return rc, bp


def decompose_alleles(row0, row1, primary, secondary, bp, refslice_size, trim_left=50, trim_right=50,
                      maxindel=1000, madc=5):
    cfg = DecompCfg(trim_left, trim_right, maxindel, madc)
    pri = C.create_string_buffer(_b(primary), len(primary) + 1)
    sec = C.create_string_buffer(_b(secondary), len(secondary) + 1)
    cap = 2 * maxindel + 4
    di = (C.c_int32 * cap)()
    de = (C.c_int32 * cap)()
    n = C.c_size_t(0)
    st = DecompStatus()
    lib().orc_decompose_alleles(C.byref(cfg), _b(row0), _b(row1), C.c_size_t(len(row0)), pri, sec,
                                C.c_size_t(len(primary)), bp, C.c_size_t(refslice_size), di, de, C.byref(n),
                                C.byref(st))
    dcp = [(di[i], de[i]) for i in range(n.value)]
    return pri.raw[:len(primary)], sec.raw[:len(secondary)], dcp, (st.kind, st.bestIns, st.bestDel, st.bestFR)


def generate_secondary_decomposed(trace, bcpos, primary, secondary):
    trace, tp = _ip(trace)
    bcpos, bp = _ip(bcpos)
    out = C.create_string_buffer(len(primary) + 1)
    lib().orc_generate_secondary_decomposed(tp, C.c_size_t(trace.shape[1]), bp, _b(primary), _b(secondary),
                                            C.c_size_t(len(primary)), out)
    return out.raw[:len(primary)]


def allelic_fraction(trace, bcpos, primary, secdecomp, trim_left=50, trim_right=50):
    trace, tp = _ip(trace)
    bcpos, bp = _ip(bcpos)
    a = C.c_double(0)
    b = C.c_double(0)
    lib().orc_allelic_fraction(tp, C.c_size_t(trace.shape[1]), bp, _b(primary), _b(secdecomp),
                               C.c_size_t(len(primary)), C.c_uint32(trim_left), C.c_uint32(trim_right),
                               C.byref(a), C.byref(b))
    return a.value, b.value


def _basecall(fn, trace, basecallpos, sigratio):
    trace, tp = _ip(trace)
    pos, pp = _ip(basecallpos)
    n = len(pos)
    pri = C.create_string_buffer(n + 1)
    sec = C.create_string_buffer(n + 1)
    con = C.create_string_buffer(n + 1)
    bc = np.zeros(max(n, 1), dtype=np.int32)
    k = fn(tp, C.c_size_t(trace.shape[1]), pp, C.c_size_t(n), C.c_float(sigratio), pri, sec, con,
           bc.ctypes.data_as(C.POINTER(C.c_int32)))
    return pri.raw[:k], sec.raw[:k], con.raw[:k], bc[:k].copy()


def basecall(trace, basecallpos, sigratio=0.33):
    return _basecall(lib().orc_basecall, trace, basecallpos, sigratio)


def ref_basecall(trace, basecallpos, sigratio=0.33):
    return _basecall(ref_lib().ref_basecall, trace, basecallpos, sigratio)


class ChainResult(C.Structure):
    _fields_ = [("score_fwd", C.c_int32), ("score_rev", C.c_int32), ("forward", C.c_int32), ("score_prelim", C.c_int32),
                ("slice_begin", C.c_uint32), ("slice_len", C.c_uint32), ("ref_pos", C.c_uint32), ("score_final", C.c_int32),
                ("btr_len", C.c_uint32), ("cells", C.c_uint64)]


def sage_chain_batch(profiles, refs, score, trim_left=50, trim_right=50, nthreads=1):
    """the `tracy align` hot section (sage.h:233-260, 311) for traces of equal shape, one trace per C thread
    (tracy_oracle_chain.c).  profiles: float32 [nt][6][mf]; refs: uint8 [nt][n].  Returns (list of dicts, total cells)."""
    profiles = np.ascontiguousarray(profiles, dtype=np.float32)
    refs = np.ascontiguousarray(refs, dtype=np.uint8)
    nt, _, mf = profiles.shape
    n = refs.shape[1]
    out = (ChainResult * max(nt, 1))()
    cap = mf + n
    btr = np.zeros((max(nt, 1), cap), dtype=np.uint8)
    sc = Score(*score)
    rc = lib().orc_sage_chain_batch(profiles.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(mf), refs.ctypes.data_as(C.c_char_p), C.c_size_t(n),
                                    C.c_uint32(nt), C.byref(sc), C.c_uint32(trim_left), C.c_uint32(trim_right), C.c_uint32(nthreads), out,
                                    btr.ctypes.data_as(C.c_char_p), C.c_size_t(cap))
    if rc != 0:
        raise MemoryError("oracle chain failed")
    res = []
    for t in range(nt):
        r = out[t]
        res.append(dict(score_fwd=r.score_fwd, score_rev=r.score_rev, forward=r.forward, score_prelim=r.score_prelim, slice_begin=r.slice_begin,
                        slice_len=r.slice_len, ref_pos=r.ref_pos, score_final=r.score_final, btr=btr[t, :r.btr_len].tobytes()))
    return res, int(sum(out[t].cells for t in range(nt)))


def gotoh_row_state(p1, p2, R, score):
    """H(R, j) and F(R, j), j = 0..n, of the semiglobal DP on the first R rows (oracle of the prefix-bound kernel)"""
    p1 = np.ascontiguousarray(p1, dtype=np.float32)
    p2 = np.ascontiguousarray(p2, dtype=np.float32)
    m, n = p1.shape[1], p2.shape[1]
    H = np.zeros(n + 1, np.int32)
    F = np.zeros(n + 1, np.int32)
    sc = Score(*score)
    lib().orc_gotoh_row_state(p1.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(m), p2.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(n),
                              C.c_size_t(R), C.byref(sc), H.ctypes.data_as(C.POINTER(C.c_int32)), F.ctypes.data_as(C.POINTER(C.c_int32)))
    return H, F
