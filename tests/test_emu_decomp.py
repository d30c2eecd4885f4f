"""decompose phase functions (tracy_amd/csrc/decompose_kernels.h) run on the host vs the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import pyoracle as orc
from decomp_cases import case_list, oracle_decompose

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(HERE, "emu", "libemu_decomp.so")
    srcs = [os.path.join(HERE, "emu", "emu_decomp.cpp"), os.path.join(ROOT, "tracy_amd/csrc/decompose_kernels.h"),
            os.path.join(ROOT, "tracy_amd/csrc/dp_lane.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, srcs[0]],
                              stderr=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.emu_phase.restype = C.c_char
    lib.emu_secdecomp.restype = C.c_uint8
    return lib


@pytest.fixture(scope="module")
def cases():
    return case_list()


def run_emu_decompose(emu, c, maxindel=1000, madc=5, bytewise=0):
    r0, r1 = c["rows"]
    pri = C.create_string_buffer(c["pri"], len(c["pri"]) + 1)
    sec = C.create_string_buffer(c["sec"], len(c["sec"]) + 1)
    di = (C.c_int32 * (2 * maxindel + 4))()
    de = (C.c_int32 * (2 * maxindel + 4))()
    out = (C.c_int32 * 6)()
    emu.emu_decompose(r0, r1, len(r0), pri, sec, len(c["pri"]), c["bp"].breakpoint, len(c["ref"]), 50, 50, maxindel, madc,
                      di, de, out, bytewise)
    n = out[4]
    return pri.raw[:len(c["pri"])], sec.raw[:len(c["sec"])], [(di[i], de[i]) for i in range(n)], tuple(out[:4])


def test_decompose_phases_match_oracle(emu, cases):
    kinds = set()
    for c in cases:
        want = oracle_decompose(c)
        got = run_emu_decompose(emu, c)
        assert got == run_emu_decompose(emu, c, bytewise=1)  # bit-set scans == byte-wise fallback
        assert got[0] == want["pri"] and got[1] == want["sec"]
        assert got[2] == want["dcp"]
        assert got[3][0] == want["status"][0]
        if want["status"][0] != 0:
            assert got[3] == want["status"]
        kinds.add(want["status"][0])
    assert 0 in kinds  # at least one plain indel shift among the synthetic traces
    # small maxindel / different MAD cut-off exercise the clamps
    c = cases[0]
    for (mi, madc) in [(1, 5), (7, 5), (40, 0), (300, 9)]:
        want = orc.decompose_alleles(c["rows"][0], c["rows"][1], c["pri"], c["sec"], c["bp"], len(c["ref"]), 50, 50, mi, madc)
        got = run_emu_decompose(emu, c, mi, madc)
        assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2] and got[3][0] == want[3][0]


def test_random_alignments_stress_the_scans(emu):
    """random gapped alignments with IUPAC secondaries, exotic reference letters, breakpoints anywhere (also never
    reached), small and large trims: every path of the scans (simple, complex, none; bit sets and byte-wise)"""
    rng = np.random.default_rng(2024)
    kinds = {}
    for it in range(60):
        nb = int(rng.integers(60, 420))
        tl, tr = (int(rng.integers(0, 30)), int(rng.integers(0, 30))) if it % 3 else (50, 50)
        if tl + tr >= nb - 5:
            tl = tr = 2
        mt = nb - tl - tr
        core = bytes(rng.choice(list(b"ACGT"), size=mt + 40).tolist())
        pri = bytearray(rng.choice(list(b"ACGT"), size=nb).tolist())
        pri[tl:tl + mt] = core[:mt]
        sec = bytearray(pri)
        shift = int(rng.integers(-12, 13))
        bpv = int(rng.integers(0, mt + 10))
        for i in range(tl + min(bpv, mt), nb - tr):
            # heterozygous tail: the secondary follows the reference shifted by `shift` (sometimes as IUPAC of both)
            src = i - tl + shift
            if 0 <= src < len(core):
                other = core[src]
                if other != pri[i]:
                    sec[i] = other if rng.random() < 0.7 else ord(orc.lib().orc_iupac2(bytes([pri[i]]), bytes([other])))
            if rng.random() < 0.03:
                sec[i] = ord("N")
        # alignment rows: the trimmed trace against a reference carrying the core plus flanks, a few gaps
        r0, r1 = bytearray(), bytearray()
        lead = int(rng.integers(0, 30))
        r0 += b"-" * lead
        r1 += bytes(rng.choice(list(b"ACGT"), size=lead).tolist())
        for i in range(mt):
            u = rng.random()
            if u < 0.01:
                r0 += b"-"; r1 += bytes([int(rng.choice(list(b"ACGT")))])
            if u > 0.99:
                r0 += bytes([pri[tl + i]]); r1 += b"-"
                continue
            r0 += bytes([pri[tl + i]])
            r1 += bytes([core[i]]) if rng.random() > 0.02 else bytes([int(rng.choice(list(b"ACGTN")))])
        trail = int(rng.integers(0, 600))
        r0 += b"-" * trail
        r1 += bytes(rng.choice(list(b"ACGT"), size=trail).tolist())
        if it % 10 == 9:
            r1[len(r1) // 2] = ord("X")  # exotic reference letter: the kernel falls back to byte-wise scans
        bp = orc.Breakpoint(1, 1, bpv, 0.5)
        c = dict(rows=(bytes(r0), bytes(r1)), pri=bytes(pri), sec=bytes(sec), bp=bp, ref=bytes(r1).replace(b"-", b""))
        mi = int(rng.choice([1000, 1000, 37, 200]))
        want = orc.decompose_alleles(c["rows"][0], c["rows"][1], c["pri"], c["sec"], bp, len(c["ref"]), tl, tr, mi, 5)
        for bw in (0, 1):
            r0b, r1b = c["rows"]
            p = C.create_string_buffer(c["pri"], nb + 1)
            s = C.create_string_buffer(c["sec"], nb + 1)
            di = (C.c_int32 * (2 * mi + 4))()
            de = (C.c_int32 * (2 * mi + 4))()
            out = (C.c_int32 * 6)()
            emu.emu_decompose(r0b, r1b, len(r0b), p, s, nb, bpv, len(c["ref"]), tl, tr, mi, 5, di, de, out, bw)
            got = (p.raw[:nb], s.raw[:nb], [(di[i], de[i]) for i in range(out[4])], tuple(out[:4]))
            assert got[0] == want[0] and got[1] == want[1], (it, bw)
            assert got[2] == want[2], (it, bw)
            assert got[3][0] == want[3][0], (it, bw)
            if want[3][0] == 1:
                assert got[3] == tuple(want[3][:4]), (it, bw, got[3], want[3])
        kinds[want[3][0]] = kinds.get(want[3][0], 0) + 1
    assert kinds.get(0, 0) > 5 and kinds.get(1, 0) > 5 and kinds.get(2, 0) >= 1, kinds


def test_complex_and_none_paths(emu):
    """hand-made alignments: identical sequences (no indel: kind 2) and a shifted tail with junk (complex)"""
    rng = np.random.default_rng(3)
    L = 400
    ref = bytes(rng.choice(list(b"ACGT"), size=L).tolist())
    pri = b"A" * 50 + ref[:300] + b"A" * 50
    sec = pri
    row0 = ref[:300] + b"-" * 100
    row1 = ref
    for bpv in (120, 300):
        bp = orc.Breakpoint(0, 1, bpv, 0.0)
        c = dict(rows=(row0, row1), pri=pri, sec=sec, bp=bp, ref=ref)
        want = orc.decompose_alleles(row0, row1, pri, sec, bp, len(ref), 50, 50, 1000, 5)
        got = run_emu_decompose(emu, c)
        assert got == run_emu_decompose(emu, c, bytewise=1)
        assert (got[0], got[1], got[2]) == (want[0], want[1], want[2]) and got[3] == want[3]
    # heterozygous tail: secondary carries the reference shifted by 3 with an extra insertion -> complex search
    tail = ref[150:300]
    sec2 = bytearray(pri)
    shifted = ref[153:303]
    sec2[50 + 150:50 + 300] = shifted
    bp = orc.Breakpoint(1, 1, 150, 0.5)
    c = dict(rows=(row0, row1), pri=pri, sec=bytes(sec2), bp=bp, ref=ref)
    want = orc.decompose_alleles(row0, row1, pri, bytes(sec2), bp, len(ref), 50, 50, 1000, 5)
    got = run_emu_decompose(emu, c)
    assert got == run_emu_decompose(emu, c, bytewise=1)
    assert (got[0], got[1], got[2]) == (want[0], want[1], want[2]) and got[3][0] == want[3][0]


def test_breakpoints_and_small_functions(emu, cases):
    for c in cases:
        p = np.ascontiguousarray(c["prof"], dtype=np.float32)
        out = (C.c_int32 * 4)()
        bd = C.c_float(0)
        emu.emu_find_breakpoint(p.ctypes.data_as(C.POINTER(C.c_float)), p.shape[1], p.shape[1], out, C.byref(bd))
        want = orc.find_breakpoint(p)
        assert (out[0], out[1], out[2]) == (want.indelshift, want.traceleft, want.breakpoint)
        assert np.float32(bd.value) == np.float32(want.bestDiff)
        r0, r1 = c["rows"]
        rc = emu.emu_homozygous(r0, r1, len(r0), out, C.byref(bd))
        wrc, wbp = orc.find_homozygous_breakpoint(r0, r1)
        assert rc == wrc and (out[0], out[1], out[2]) == (wbp.indelshift, wbp.traceleft, wbp.breakpoint)
        assert np.float32(bd.value) == np.float32(wbp.bestDiff)
    # the column-mask formulation of findHomozygousBreakpoint (what homozygous_kernel runs) on rows of every awkward shape
    from decomp_cases import random_row_pairs
    seen = set()
    for r0, r1 in random_row_pairs(5, 600):
        rc = emu.emu_homozygous(r0, r1, len(r0), out, C.byref(bd))
        wrc, wbp = orc.find_homozygous_breakpoint(r0, r1)
        assert rc == wrc
        seen.add((rc, wbp.indelshift if rc == 1 else 0))
        if rc == 1:
            assert (out[0], out[1], out[2]) == (wbp.indelshift, wbp.traceleft, wbp.breakpoint)
            assert np.float32(bd.value) == np.float32(wbp.bestDiff)
    assert seen == {(0, 0), (-1, 0), (1, 0), (1, 1)}
    # degenerate alignments: the two failure codes
    out = (C.c_int32 * 4)()
    bd = C.c_float(0)
    assert emu.emu_homozygous(b"----", b"ACGT", 4, out, C.byref(bd)) == orc.find_homozygous_breakpoint(b"----", b"ACGT")[0] == 0
    short = b"ACGTACGTAC" * 3
    assert emu.emu_homozygous(short, short, len(short), out, C.byref(bd)) == orc.find_homozygous_breakpoint(short, short)[0] == -1
    # phaseRefAllele over the whole alphabet
    lib = orc.lib()
    for p in b"ACGTN":
        for s in b"ACGTNRYSWKMX":
            for r in b"ACGTN-X":
                pri = bytes([p]); sec = bytes([s])
                want = orc.decompose_alleles(bytes([p]) , bytes([r]), pri, sec, orc.Breakpoint(1, 1, 5, 1.0), 10, 0, 0, 1, 5)
                got = emu.emu_phase(C.c_char(bytes([p])), C.c_char(bytes([s])), C.c_char(bytes([r])))
                # the walk applies the phase when row1 != primary: compare through its effect
                if r != p:
                    exp_sec = want[1]
                    if got != b"N":
                        assert want[0] == bytes([r]) and exp_sec == got
                    else:
                        assert want[0] == pri and exp_sec == sec


def test_traverse_of_the_whole_alignment_rewrites_every_lane_segment(emu):
    """"No InDel detected, traverse the whole alignment" (decompose.h:327-343) with basecalls to phase all along the alignment: every
    lane's segment starts at its own basecall.  (A scoring with cheap mismatches aligns heterozygous traces so that the shift scans
    find nothing and the traverse has dozens of positions to rewrite; with tracy's default scoring it rarely has any, which hid a
    segment-offset table that the pick phases had overwritten.)"""
    from tracy_amd import hostlib
    sc = (1, -1, -2, -1)
    d = hostlib.synth_decompose_batch(12345, 8, 2200, 700, 0, mix=1)
    traversed = 0
    for i in range(8):
        sig, pos = d["signal"][i], d["bcpos"][i]
        pri, sec = d["primary"][i].tobytes(), d["secondary"][i].tobytes()
        prof = orc.create_profile_trace(sig, pos, pri, sec, 50, 50)
        ref = d["refs"][i].tobytes()
        bp = orc.find_breakpoint(prof)
        fwd = orc.create_profile_str(ref)
        rev = orc.revcomp_profile(fwd)
        use = fwd if orc.gotoh_score_prof(prof, fwd, 1, 0, sc) > orc.gotoh_score_prof(prof, rev, 1, 0, sc) else rev
        _, btr = orc.gotoh_prof(prof, use, 1, 0, sc)
        r0, r1 = orc.create_alignment_prof(btr, prof, use)
        c = dict(ref=ref, pri=pri, sec=sec, bp=bp, rows=(r0, r1))
        want = orc.decompose_alleles(r0, r1, pri, sec, bp, len(ref), 50, 50, 1000, 5)
        got = run_emu_decompose(emu, c)
        assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2] and got[3][0] == want[3][0], i
        if want[3][0] == 2 and sum(a != b for a, b in zip(pri, want[0])) > 20:
            traversed += 1
    assert traversed >= 3
