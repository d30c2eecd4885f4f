"""ctypes binding of libtracy_msa.so: the progressive multiple alignment of `tracy assemble` (tracy_amd/host/msa.hpp)
with its dynamic programs on the GPU through the C ABI."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        capi.lib()  # libtracy_hip.so first
        p = os.path.join(_HERE, "lib", "libtracy_msa.so")
        if not os.path.exists(p):
            raise ImportError("tracy_amd: %s is missing -- run `python tracy_amd/build.py`" % p)
        _LIB = C.CDLL(p)
        _LIB.tracymsa_msa.restype = C.c_int64
        _LIB.tracymsa_msa_group.restype = C.c_int64
        _LIB.tracymsa_pair_list.restype = C.c_uint64
        _LIB.tracymsa_consensus.restype = C.c_int64
    return _LIB


def _pack(profiles):
    lens = np.array([p.shape[1] for p in profiles], dtype=np.uint32)
    offs = np.zeros(max(len(profiles), 1), dtype=np.uint64)
    if len(profiles):
        offs[1:len(profiles)] = np.cumsum(6 * lens.astype(np.uint64))[:-1]
    data = np.concatenate([np.ascontiguousarray(p, dtype=np.float32).ravel() for p in profiles]) if profiles else np.zeros(1, np.float32)
    return data, offs, lens


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def pair_list(n):
    """the (i1, i2) index arrays msa() hands to the device for its distance matrix (msa.h:33-42)"""
    cnt = n * (n - 1) // 2
    i1 = np.zeros(max(cnt, 1), np.uint32)
    i2 = np.zeros(max(cnt, 1), np.uint32)
    got = lib().tracymsa_pair_list(C.c_uint32(n), _ptr(i1, C.c_uint32), _ptr(i2, C.c_uint32))
    assert got == cnt
    return i1[:cnt], i2[:cnt]


def msa(ctx, profiles, score, group=None):
    """msa() of msa.h:326-368 -> (rows: list of bytes, seqidx: list of int); group: spread the distance matrix over a device group"""
    data, offs, lens = _pack(profiles)
    n = len(profiles)
    cap = int(lens.sum()) * max(n, 1) + 16
    rows = C.create_string_buffer(cap)
    sidx = np.zeros(max(n, 1), np.uint32)
    nrows = C.c_uint32(0)
    prm = capi.Params(score[0], score[1], score[2], score[3], 1, 1)
    ncol = lib().tracymsa_msa_group(ctx._h, group._g if group is not None else None, C.byref(prm), _ptr(data, C.c_float), _ptr(offs, C.c_uint64),
                                    _ptr(lens, C.c_uint32), C.c_uint32(n), rows, C.c_uint64(cap), _ptr(sidx, C.c_uint32), C.byref(nrows))
    if ncol < 0:
        raise RuntimeError("msa failed: %s" % capi.lib().tracyhip_last_error().decode())
    return [rows.raw[i * ncol:(i + 1) * ncol] for i in range(nrows.value)], sidx[:nrows.value].tolist()


def consensus(rows, fraction_called=0.5, ignore_last=False):
    """consensus() of msa.h:165-254 -> (gapped, cs, qstr)"""
    nrows, ncol = len(rows), len(rows[0]) if rows else 0
    blob = b"".join(rows) + b"\0"
    g, c, q = (C.create_string_buffer(ncol + 1) for _ in range(3))
    n = lib().tracymsa_consensus(C.c_float(fraction_called), blob, C.c_uint32(nrows), C.c_uint64(ncol), int(ignore_last), g, c, q)
    return g.raw[:ncol], c.raw[:n], q.raw[:n]


def profile_of_alignment(rows):
    nrows, ncol = len(rows), len(rows[0]) if rows else 0
    out = np.zeros((6, max(ncol, 1)), np.float32)
    lib().tracymsa_profile_of_alignment(b"".join(rows) + b"\0", C.c_uint32(nrows), C.c_uint64(ncol), _ptr(out, C.c_float))
    return out[:, :ncol]


def rev_seq_based_on_dist(ctx, profiles, score):
    """revSeqBasedOnDist() of msa.h:258-323 -> (profiles after flipping, fwd flags)"""
    data, offs, lens = _pack(profiles)
    n = len(profiles)
    fwd = np.ones(max(n, 1), np.uint8)
    prm = capi.Params(score[0], score[1], score[2], score[3], 1, 1)
    rc = lib().tracymsa_rev_seq(ctx._h, C.byref(prm), _ptr(data, C.c_float), _ptr(offs, C.c_uint64), _ptr(lens, C.c_uint32), C.c_uint32(n),
                                _ptr(fwd, C.c_uint8))
    if rc != 0:
        raise RuntimeError("revSeqBasedOnDist failed: %s" % capi.lib().tracyhip_last_error().decode())
    out = [data[int(offs[i]):int(offs[i]) + 6 * int(lens[i])].reshape(6, int(lens[i])).copy() for i in range(n)]
    return out, fwd[:n].astype(bool).tolist()
