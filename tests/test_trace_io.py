"""Chromatogram file formats either side of the hot path: the host
ABIF/SCF readers, the quality estimate that ends basecall() and traceTxtOut, checked against the REFERENCE's
own abif.h (oracle/_ref, this container only) and against committed golden vectors generated from it
(tests/golden/make_trace_io_golden.py).  SCF has no compilable reference (scf.h needs Boost): the SCF test
checks the documented v3 layout only ("parity unpinned")."""
import base64
import json
import os
import struct

import numpy as np
import pytest

import pyoracle as orc
from test_host_and_abi import make_trace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "trace_io.json")


def ref_read_trace(path):
    from tracy_amd import hostlib
    ref = orc.ref_lib()
    return hostlib._read_trace_with(ref, "ref_", path, False)


def same_trace(a, b):
    return (np.array_equal(a["signal"], b["signal"]) and np.array_equal(a["basecallpos"], b["basecallpos"]) and
            a["basecalls1"] == b["basecalls1"] and a["basecalls2"] == b["basecalls2"] and np.array_equal(a["qual"], b["qual"]))


def abif_cases(rng, tmp):
    """(path, description) of ABIF files written by the build's writer, covering the reader's branches"""
    from tracy_amd import hostlib
    out = []
    for it, (nb, order, het, sec) in enumerate([(40, b"GATC", 0.0, False), (120, b"ACGT", 0.4, True), (7, b"TGCA", 0.0, False),
                                                (2, b"CATG", 0.0, True), (300, b"GATC", 0.2, False)]):
        tr, pos = make_trace(rng, nb, het=het)
        tr = np.minimum(tr, 32000)
        if it == 1:
            tr[1, 50:60] = -7  # negative samples are sign-extended
        pri = bytes(rng.choice(list(b"ACGTNRY"), size=nb).tolist())
        sec_s = bytes(rng.choice(list(b"ACGTK"), size=nb - (1 if it == 3 else 0)).tolist()) if sec else b""
        qual = rng.integers(0, 62, size=nb + (3 if it == 4 else 0)).astype(np.uint8)  # more qualities than calls: cut
        p = os.path.join(tmp, "case%d.ab1" % it)
        hostlib.write_abif(p, tr, pos, pri, qual, sec_s, order)
        out.append(p)
    return out


@pytest.mark.skipif(orc.ref_lib() is None, reason="oracle/_ref exists only where /root/reference does")
def test_readab_matches_the_reference(tmp_path):
    from tracy_amd import hostlib
    rng = np.random.default_rng(41)
    for p in abif_cases(rng, str(tmp_path)):
        got, want = hostlib.read_trace(p), ref_read_trace(p)
        assert got is not None and want is not None and got["format"] == 0
        assert same_trace(got, want), p
        assert got["signal"].shape[1] > 0 and len(got["basecallpos"]) > 0
    # a file without basecalls is refused by both
    tr, pos = make_trace(rng, 10)
    p = str(tmp_path / "empty.ab1")
    hostlib.write_abif(p, tr, pos, b"", np.zeros(0, np.uint8))
    assert hostlib.read_trace(p) is None and ref_read_trace(p) is None
    # not ABIF
    p = str(tmp_path / "junk.ab1")
    open(p, "wb").write(b"JUNK" + bytes(200))
    assert hostlib.read_trace(p) is None and ref_read_trace(p) is None


def test_readab_golden_files():
    """ABIF files (base64) + what the reference's readab() returned for them"""
    from tracy_amd import hostlib
    import tempfile
    cases = json.load(open(GOLD))["abif"]
    assert len(cases) >= 4
    with tempfile.TemporaryDirectory() as tmp:
        for i, c in enumerate(cases):
            p = os.path.join(tmp, "g%d.ab1" % i)
            open(p, "wb").write(base64.b64decode(c["file_b64"]))
            got = hostlib.read_trace(p)
            assert got["signal"].tolist() == c["signal"] and got["basecallpos"].tolist() == c["basecallpos"]
            assert got["basecalls1"] == base64.b64decode(c["basecalls1_b64"]) and got["basecalls2"] == base64.b64decode(c["basecalls2_b64"])
            assert got["qual"].tolist() == c["qual"]


def test_writer_round_trip_and_dye_order(tmp_path):
    from tracy_amd import hostlib
    rng = np.random.default_rng(5)
    tr, pos = make_trace(rng, 64, het=0.3)
    pri = bytes(rng.choice(list(b"ACGT"), size=64).tolist())
    for order in (b"GATC", b"ACGT", b"TCGA"):
        p = str(tmp_path / ("o_%s.ab1" % order.decode()))
        hostlib.write_abif(p, tr, pos, pri, np.full(64, 30, np.uint8), order=order)
        got = hostlib.read_trace(p)
        assert np.array_equal(got["signal"], tr) and np.array_equal(got["basecallpos"], pos) and got["basecalls1"] == pri
        assert got["basecalls2"] == bytes(64)  # no P2BA tag: NUL padded like the reference's resize()
    # truncated file: refused, never read out of bounds
    raw = open(p, "rb").read()
    q = str(tmp_path / "cut.ab1")
    open(q, "wb").write(raw[:len(raw) // 2])
    assert hostlib.read_trace(q) is None


def estqual_cases(rng):
    for it in range(10):
        nb = [1, 4, 9, 11, 25, 60, 150, 400, 33, 12][it]
        yield make_trace(rng, nb, het=[0.0, 0.3, 0.8][it % 3])


@pytest.mark.skipif(orc.ref_lib() is None, reason="oracle/_ref exists only where /root/reference does")
def test_estimated_qualities_and_trace_txt_match_the_reference(tmp_path):
    from tracy_amd import hostlib
    import ctypes as C
    ref = orc.ref_lib()
    ref.ref_basecall_qual.restype = C.c_size_t
    rng = np.random.default_rng(77)
    for tr, pos in estqual_cases(rng):
        got = hostlib.basecall_qual(tr, pos, 0.33)
        n = len(pos)
        pri, sec, con = (C.create_string_buffer(n + 1) for _ in range(3))
        bc = np.zeros(max(n, 1), np.int32)
        q = np.zeros(max(n, 1), np.uint8)
        k = ref.ref_basecall_qual(tr.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(tr.shape[1]), pos.ctypes.data_as(C.POINTER(C.c_int32)),
                                  C.c_size_t(n), C.c_float(0.33), pri, sec, con, bc.ctypes.data_as(C.POINTER(C.c_int32)),
                                  q.ctypes.data_as(C.POINTER(C.c_uint8)))
        assert k == len(got[0]) and got[0] == pri.raw[:k] and got[1] == sec.raw[:k]
        assert np.array_equal(got[4], q[:k]), (n, got[4].tolist(), q[:k].tolist())
        a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
        for (lt, rt) in [(0, 0), (3, 5), (1000, 2)]:
            ra = hostlib.trace_txt(a, tr, pos, 0.33, lt, rt)
            rb = ref.ref_trace_txt(os.fsencode(b), tr.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(tr.shape[1]),
                                   pos.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(n), C.c_float(0.33), C.c_uint32(lt), C.c_uint32(rt))
            assert ra == rb
            if ra == 0:
                assert open(a, "rb").read() == open(b, "rb").read()


def test_estimated_qualities_golden():
    from tracy_amd import hostlib
    cases = json.load(open(GOLD))["estqual"]
    assert len(cases) >= 6
    for c in cases:
        got = hostlib.basecall_qual(np.array(c["trace"], np.int32), np.array(c["basecallpos"], np.int32), c["sigratio"])
        assert got[4].tolist() == c["estQual"] and got[0].decode() == c["primary"]


def write_scf3(path, tr, pos, version=b"3.00"):
    """minimal SCF v3 per the published layout: 128-byte header, per-channel second differences (uint16 BE)"""
    ns, nb = tr.shape[1], len(pos)
    samples_at, bases_at = 128, 128 + 8 * ns
    head = struct.pack(">4s8I4sI", b".scf", ns, samples_at, nb, 0, 0, bases_at, 0, 0, version, 2)
    head = head.ljust(128, b"\0")
    body = b""
    for c in range(4):
        d = tr[c].astype(np.int64)
        for _ in range(2):
            d = np.diff(np.concatenate([[0], d]))
        body += (d & 0xFFFF).astype(">u2").tobytes()
    body += pos.astype(">i4").tobytes()
    open(path, "wb").write(head + body)


def test_readscf_v3_layout(tmp_path):
    from tracy_amd import hostlib
    rng = np.random.default_rng(3)
    tr, pos = make_trace(rng, 50, het=0.2)
    p = str(tmp_path / "x.scf")
    write_scf3(p, tr, pos)
    got = hostlib.read_trace(p)
    assert got["format"] == 1 and np.array_equal(got["signal"], tr) and np.array_equal(got["basecallpos"], pos)
    assert got["qual"].tolist() == [0] * 50
    write_scf3(p, tr, pos, version=b"2.00")  # the reference refuses SCF < 3.0 (scf.h:91-94)
    assert hostlib.read_trace(p) is None
