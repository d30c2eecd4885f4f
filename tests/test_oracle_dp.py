"""Oracle (C restatement) vs the independent full-matrix formulation + hand-derived known answers."""
import numpy as np
import pytest

import npref
import pyoracle as orc

SCORES = [(3, -5, -10, -4), (5, -4, -10, -1)]
CONFIGS = [(0, 0), (1, 0), (0, 1), (1, 1)]


def rand_seq(rng, n, alpha):
    return bytes(rng.choice(list(alpha), size=n).tolist())


def rand_profile(rng, n, sharp=False):
    p = np.zeros((6, n), dtype=np.float32)
    x = rng.random((4, n)).astype(np.float32)
    if sharp:
        x = x ** 8
    p[:4] = x / x.sum(axis=0, keepdims=True)
    return p


@pytest.mark.parametrize("sc", SCORES)
@pytest.mark.parametrize("cfg", CONFIGS)
def test_gotoh_str_vs_fullmatrix(sc, cfg):
    rng = np.random.default_rng(hash((sc, cfg)) % 2**32)
    for it in range(120):
        alpha = [b"A", b"AC", b"ACG", b"ACGT", b"ACGTN"][it % 5]
        m, n = int(rng.integers(0, 24)), int(rng.integers(0, 30))
        s1, s2 = rand_seq(rng, m, alpha), rand_seq(rng, n, alpha)
        want = npref.gotoh_full(m, n, lambda r, c: npref.sub_str(s1, s2, r, c, sc), cfg[0], cfg[1], sc)
        got = orc.gotoh_str(s1, s2, cfg[0], cfg[1], sc)
        assert got == want, (s1, s2, cfg, sc)
        assert orc.gotoh_score_str(s1, s2, cfg[0], cfg[1], sc) == want[0]


@pytest.mark.parametrize("cfg", CONFIGS)
def test_gotoh_profile_vs_fullmatrix(cfg):
    sc = SCORES[0]
    rng = np.random.default_rng(7 + cfg[0] * 2 + cfg[1])
    for it in range(30):
        m, n = int(rng.integers(1, 14)), int(rng.integers(1, 18))
        p1 = rand_profile(rng, m, sharp=it % 2 == 0)
        p2 = rand_profile(rng, n) if it % 3 else orc.create_profile_str(rand_seq(rng, n, b"ACGTN-x"))
        want = npref.gotoh_full(m, n, lambda r, c: npref.sub_prof(p1, p2, r, c, sc), cfg[0], cfg[1], sc)
        assert orc.gotoh_prof(p1, p2, cfg[0], cfg[1], sc) == want
        assert orc.gotoh_score_prof(p1, p2, cfg[0], cfg[1], sc) == want[0]


@pytest.mark.parametrize("cfg", CONFIGS)
def test_needle_vs_fullmatrix(cfg):
    sc = SCORES[1]
    rng = np.random.default_rng(11 + cfg[0] * 2 + cfg[1])
    for it in range(80):
        alpha = [b"A", b"AC", b"ACGT"][it % 3]
        m, n = int(rng.integers(0, 20)), int(rng.integers(0, 24))
        s1, s2 = rand_seq(rng, m, alpha), rand_seq(rng, n, alpha)
        want = npref.needle_full(m, n, lambda r, c: npref.sub_str(s1, s2, r, c, sc), cfg[0], cfg[1], sc)
        assert orc.needle_str(s1, s2, cfg[0], cfg[1], sc) == want
        assert orc.needle_score_str(s1, s2, cfg[0], cfg[1], sc) == want[0]
    for it in range(10):
        m, n = int(rng.integers(1, 10)), int(rng.integers(1, 12))
        p1, p2 = rand_profile(rng, m), rand_profile(rng, n)
        want = npref.needle_full(m, n, lambda r, c: npref.sub_prof_double(p1, p2, r, c, sc), cfg[0], cfg[1], sc)
        assert orc.needle_prof(p1, p2, cfg[0], cfg[1], sc) == want
        assert orc.needle_score_prof(p1, p2, cfg[0], cfg[1], sc) == want[0]


def test_known_answers():
    sc = (3, -5, -10, -4)
    # identical strings, global: m matches
    assert orc.gotoh_str(b"ACGT", b"ACGT", 0, 0, sc) == (12, b"ssss")
    # one base deleted from the trace, global: 3 matches + one gap of length 1 = go + ge
    s, btr = orc.gotoh_str(b"ACT", b"ACGT", 0, 0, sc)
    assert s == 9 - 14 and sorted(btr) == sorted(b"sssh")
    # degenerate sizes run only the init row / column (gotoh.h:106-123)
    assert orc.gotoh_str(b"", b"ACG", 0, 0, sc) == (-10 - 12, b"hhh")
    assert orc.gotoh_str(b"", b"ACG", 1, 0, sc) == (0, b"hhh")
    assert orc.gotoh_str(b"AC", b"", 0, 0, sc) == (-10 - 8, b"vv")
    assert orc.gotoh_str(b"AC", b"", 0, 1, sc) == (0, b"vv")
    assert orc.gotoh_str(b"", b"", 1, 1, sc) == (0, b"")
    # semiglobal: trace inside the reference costs nothing at the flanks; alignment length = n
    s, btr = orc.gotoh_str(b"GATTACA", b"CCCCGATTACATTTT", 1, 0, sc)
    assert s == 21 and btr == b"hhhh" + b"s" * 7 + b"hhhh"
    # raw byte compare: N matches N, lower case never matches upper case (align.h:100)
    assert orc.gotoh_score_str(b"N", b"N", 0, 0, sc) == 3
    assert orc.gotoh_score_str(b"a", b"A", 0, 0, sc) == -5
    # tie order h > v > diag: with zero-cost everything the walk takes h first (from the end)
    assert orc.gotoh_str(b"A", b"C", 0, 0, (0, 0, 0, 0)) [1] in (b"hv", b"vh", b"s")


def test_alignment_rows_and_trim():
    sc = (3, -5, -10, -4)
    s1, s2 = b"GATTACA", b"CCCCGATTACATTTT"
    _, btr = orc.gotoh_str(s1, s2, 1, 0, sc)
    r0, r1 = orc.create_alignment_str(btr, s1, s2)
    assert r0 == b"----GATTACA----" and r1 == s2
    # trimReferenceSlice: ri=4, risize=7; trims of 2: ri>=2 -> ri=2,risize=9; 2+9+2 < 15 -> risize=11
    assert orc.trim_reference_slice(r0, r1, 2, 2, len(s2), True)[:3] == (2, 11, 2)
    assert orc.trim_reference_slice(r0, r1, 5, 5, len(s2), True)[:3] == (4, 7, 4)
    assert orc.trim_reference_slice(r0, r1, 2, 2, len(s2), False)[:3] == (2, 11, 2)
    # profile consensus chars: argmax, first max wins, rows 4/5 -> 'N', all-zero column -> 'A'
    p = orc.create_profile_str(b"ACGTN-x")
    rows = orc.create_alignment_prof(b"s" * 7, p, p)
    assert rows[0] == b"ACGTNNA"


def test_profile_helpers():
    p = orc.create_profile_str(b"AcgTn-z")
    assert p[:, 0].tolist() == [1, 0, 0, 0, 0, 0] and p[:, 5].tolist() == [0, 0, 0, 0, 0, 1]
    assert p[:, 6].sum() == 0
    rc = orc.revcomp_profile(p)
    assert rc[3, 6] == 1 and rc[4, 2] == 1 and rc[5, 1] == 1 and rc[0, 3] == 1
    # createProfile(Trace, BaseCalls): float/double mix of profile.h:46-49
    trace = np.array([[900, 10, 0], [30, 500, 0], [20, 480, 0], [50, 10, 0]], dtype=np.int32)
    prof = orc.create_profile_trace(trace, [0, 1, 2], b"ACN", b"ASN")
    tot = np.float32(900)
    allb = np.float32(1000)
    nf = np.float32(tot / allb)
    want0 = np.float32(float(np.float32(nf * np.float32(np.float32(900) / tot))) + float(np.float32(1) - nf) * 0.25)
    assert prof[0, 0] == want0
    assert prof[:4, 2].tolist() == [0.25] * 4  # totalsig == 0
    assert prof[4:].sum() == 0
    # column 1: C primary, S secondary -> C and G called
    assert prof[0, 1] < prof[1, 1] and prof[3, 1] < prof[2, 1]


def test_c_chain_matches_the_python_composition():
    """oracle/tracy_oracle_chain.c (the CPU baseline of bench.py, one trace per pthread) == the chain composed in Python
    from the same oracle functions (tests/sage_oracle.py), forward and reverse traces, odd trims"""
    import sage_oracle as so
    from tracy_amd import hostlib
    refs, profs, rev = hostlib.synth_align(700, 6, 1500, 300, 2)
    for (tl, tr) in [(50, 50), (0, 7), (200, 200)]:
        res, cells = orc.sage_chain_batch(profs, refs, (3, -5, -10, -4), tl, tr, 3)
        assert cells > 0 and len(set(r["forward"] for r in res)) == 2
        for i in range(6):
            w = so.align_trace(profs[i], refs[i].tobytes(), (3, -5, -10, -4), tl, tr)
            for k in ("score_fwd", "score_rev", "forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
                assert int(res[i][k]) == int(w[k]), (i, k)
            assert res[i]["btr"] == w["btr"]
