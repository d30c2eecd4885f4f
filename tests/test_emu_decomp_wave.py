"""decomp_wave_body (tracy_amd/csrc/decompose_wave.h: decomposeAlleles as one wave per trace, working set in LDS) on the 64-fiber host
wave vs the oracle and vs the step-wise phase functions (decompose_kernels.h) it replaces on the fast path."""
import ctypes as C

import numpy as np
import pytest

import pyoracle as orc
from decomp_cases import case_list, oracle_decompose
from test_emu_decomp import emu  # noqa: F401  (the fixture builds tests/emu/libemu_decomp.so)


def run_wave(emu, r0, r1, pri, sec, bpv, reflen, tl=50, tr=50, maxindel=1000, madc=5, caps=None):
    nb = len(pri)
    p = C.create_string_buffer(pri, nb + 1)
    s = C.create_string_buffer(sec, nb + 1)
    di = (C.c_int32 * (2 * maxindel + 4))()
    de = (C.c_int32 * (2 * maxindel + 4))()
    out = (C.c_int32 * 6)()
    if caps is None:  # as launch_decompose provisions: basecalls, maxindel, the reference-row span they can reach
        capB = min(2048, (nb + 63) // 64 * 64 or 64)
        capI = min(1024, maxindel)
        caps = ((capB + capI + 256 + 63) // 64 * 64, capB, capI, min(capI, capB // 2 + 1))
    todo = emu.emu_decompose_wave(r0, r1, len(r0), p, s, nb, bpv, reflen, tl, tr, maxindel, madc, di, de, out, *caps)
    return todo, (p.raw[:nb], s.raw[:nb], [(di[i], de[i]) for i in range(out[4])], tuple(out[:4])), out[5]


def check(got, want):
    assert got[0] == want[0] and got[1] == want[1]
    assert got[2] == want[2]
    assert got[3][0] == want[3][0]
    if want[3][0] == 1:
        assert got[3] == tuple(want[3][:4])


def test_wave_body_matches_oracle_on_the_synthetic_traces(emu):  # noqa: F811
    kinds = set()
    for c in case_list():
        w = oracle_decompose(c)
        want = (w["pri"], w["sec"], w["dcp"], w["status"])
        todo, got, lds = run_wave(emu, c["rows"][0], c["rows"][1], c["pri"], c["sec"], c["bp"].breakpoint, len(c["ref"]))
        assert todo == 0
        check(got, want)
        kinds.add(want[3][0])
        # the same trace with LDS provisioned as the launcher would for a 1000-basecall batch: 12-13 KB
        assert lds < 22 * 1024
    assert 0 in kinds
    c = case_list()[0]
    for (mi, madc) in [(1, 5), (7, 5), (40, 0), (300, 9)]:
        want = orc.decompose_alleles(c["rows"][0], c["rows"][1], c["pri"], c["sec"], c["bp"], len(c["ref"]), 50, 50, mi, madc)
        todo, got, _ = run_wave(emu, c["rows"][0], c["rows"][1], c["pri"], c["sec"], c["bp"].breakpoint, len(c["ref"]), 50, 50, mi, madc)
        assert todo == 0
        check(got, want)


def test_wave_body_random_alignments(emu):  # noqa: F811
    """the generator of test_emu_decomp.test_random_alignments_stress_the_scans: IUPAC secondaries, exotic reference letters, breakpoints
    anywhere (also never reached), small and large trims -- simple, complex and none; bit sets and byte-wise"""
    rng = np.random.default_rng(2025)
    kinds = {}
    unprovisioned = 0
    for it in range(90):
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
            src = i - tl + shift
            if 0 <= src < len(core):
                other = core[src]
                if other != pri[i]:
                    sec[i] = other if rng.random() < 0.7 else ord(orc.lib().orc_iupac2(bytes([pri[i]]), bytes([other])))
            if rng.random() < 0.03:
                sec[i] = ord("N")
            if rng.random() < 0.01:
                pri[i] = ord("N")
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
            r1[len(r1) // 2] = ord("X")
        bp = orc.Breakpoint(1, 1, bpv, 0.5)
        ref = bytes(r1).replace(b"-", b"")
        mi = int(rng.choice([1000, 1000, 37, 200]))
        want = orc.decompose_alleles(bytes(r0), bytes(r1), bytes(pri), bytes(sec), bp, len(ref), tl, tr, mi, 5)
        todo, got, _ = run_wave(emu, bytes(r0), bytes(r1), bytes(pri), bytes(sec), bpv, len(ref), tl, tr, mi, 5)
        if todo == 1:
            # a breakpoint behind the trimmed trace wraps maxins (decompose.h:250): maxindel insertion shifts, more than a trace of this
            # length is provisioned for -- left to decompose_kernel, basecalls untouched
            maxins = (nb - (tr + bpv + tl)) & 0xffffffff
            assert min(mi, maxins // 2) > min(min(1024, mi), ((nb + 63) // 64 * 64) // 2 + 1), it
            assert got[0] == bytes(pri) and got[1] == bytes(sec)
            unprovisioned += 1
            continue
        assert todo == 0, it
        check(got, want)
        kinds[want[3][0]] = kinds.get(want[3][0], 0) + 1
    assert kinds.get(0, 0) > 5 and kinds.get(1, 0) > 5 and kinds.get(2, 0) >= 1, kinds
    assert unprovisioned < 10


def test_wave_body_complex_none_and_unprovisioned(emu):  # noqa: F811
    rng = np.random.default_rng(3)
    L = 400
    ref = bytes(rng.choice(list(b"ACGT"), size=L).tolist())
    pri = b"A" * 50 + ref[:300] + b"A" * 50
    row0 = ref[:300] + b"-" * 100
    row1 = ref
    for bpv in (120, 300, 0, 5000):
        bp = orc.Breakpoint(0, 1, bpv, 0.0)
        want = orc.decompose_alleles(row0, row1, pri, pri, bp, len(ref), 50, 50, 1000, 5)
        todo, got, _ = run_wave(emu, row0, row1, pri, pri, bpv, len(ref))
        if bpv == 5000:  # (a breakpoint behind the trace wraps maxins: maxindel insertion shifts, not provisioned)
            assert todo == 1 and got[0] == pri
            continue
        assert todo == 0
        check(got, want)
        assert got[3] == tuple(want[3][:4])
    sec2 = bytearray(pri)
    sec2[50 + 150:50 + 300] = ref[153:303]
    bp = orc.Breakpoint(1, 1, 150, 0.5)
    want = orc.decompose_alleles(row0, row1, pri, bytes(sec2), bp, len(ref), 50, 50, 1000, 5)
    todo, got, _ = run_wave(emu, row0, row1, pri, bytes(sec2), 150, len(ref))
    assert todo == 0
    check(got, want)
    # not provisioned: alignment longer than the LDS rows, more basecalls than the LDS strings, maxindel beyond the tables, trims outside
    # the trace -> left to decompose_kernel (to-do word 1), basecalls untouched
    for caps, tl, tr, mi in (((128, 448, 1000, 225), 50, 50, 1000), ((1728, 384, 1000, 225), 50, 50, 1000), ((1728, 448, 512, 225), 50, 50, 1000),
                             ((1728, 448, 1000, 225), 50, 401, 1000), ((1728, 448, 1000, 225), -1, 50, 1000), ((1728, 448, 1000, 20), 50, 50, 1000)):
        todo, got, _ = run_wave(emu, row0, row1, pri, bytes(sec2), 150, len(ref), tl, tr, mi, 5, caps=caps)
        assert todo == 1 and got[0] == pri and got[1] == bytes(sec2), (caps, tl, tr)
    # a trace row with more bases than basecalls behind the left trim
    todo, got, _ = run_wave(emu, row0, row1, pri[:320], pri[:320], 150, len(ref))
    assert todo == 1 and got[0] == pri[:320]


def test_wave_traverse_of_the_whole_alignment(emu):  # noqa: F811
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
        want = orc.decompose_alleles(r0, r1, pri, sec, bp, len(ref), 50, 50, 1000, 5)
        todo, got, _ = run_wave(emu, r0, r1, pri, sec, bp.breakpoint, len(ref))
        assert todo == 0
        check(got, want)
        if want[3][0] == 2 and sum(a != b for a, b in zip(pri, want[0])) > 20:
            traversed += 1
    assert traversed >= 3


def test_wave_body_whole_window_alignments(emu):  # noqa: F811
    """alignments as `tracy decompose` makes them -- the trimmed trace against a whole reference window, up to 4096 columns with 1-2 kb of
    gap columns either side: the gap bits of sixteen 256-column chunks put into column order, only the reachable span of the reference row staged"""
    from test_gpu_decompose import _random_decomp_cases
    done = 0
    for c in _random_decomp_cases(99, 12, long_window=True):
        want = orc.decompose_alleles(c["rows"][0], c["rows"][1], c["pri"], c["sec"], c["bp"], c["reflen"], c["tl"], c["tr"], 1000, 5)
        todo, got, lds = run_wave(emu, c["rows"][0], c["rows"][1], c["pri"], c["sec"], c["bp"].breakpoint, c["reflen"], c["tl"], c["tr"])
        assert lds < 14 * 1024
        if todo == 1:
            assert got[0] == c["pri"] and got[1] == c["sec"]
            continue
        check(got, want)
        done += 1
    assert done >= 9
