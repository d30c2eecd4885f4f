"""The band kernels (tracy_amd/csrc/band16.h: four pairs per wave, sixteen lanes per pair, strips handed round the lanes of a
DPP row) on the host emulator against the oracle: traceback strings / scores / the two ends of the origin-tracking sweep on
bands that hold every optimal path are those of the whole matrix (gotoh.h:71-175); the certificate the pipelines use for it."""
import os
import random
import sys

import numpy as np

import pyoracle as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

SC = (3, -5, -10, -4)
COMP = bytes.maketrans(b"ACGT", b"TGCA")


def mutate(s, rate, rng):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice(b"ACGT"))
            out.append(ch)
        elif x < rate:
            out.append(rng.choice(b"ACGTN"))
        else:
            out.append(ch)
    return bytes(out) or b"A"


def ends_of(btr, n):
    fwd = btr[::-1]
    lead = len(fwd) - len(fwd.lstrip(b"h"))
    trail = len(fwd) - len(fwd.rstrip(b"h"))
    return lead, n - trail


def gap_budget(score, m, hfree, sc=SC):
    """most interior gap steps an alignment of that score can hold: every row gives at most `match`, every gap step costs |ge|"""
    return max(0, (sc[0] * m - score)) // (-sc[3])


def test_band16_traceback_strings_equals_whole_matrix_when_certified():
    rng = random.Random(11)
    checked = certified = 0
    for it in range(90):
        K = rng.choice([4, 8, 12])
        hfree = rng.choice([0, 1])
        pairs, wants = [], []
        for q in range(rng.randint(1, 4)):
            m = rng.randint(1, 200 if it % 3 else 40)
            a = bytes(rng.choice(b"ACGTN") if rng.random() < 0.03 else rng.choice(b"ACGT") for _ in range(m))
            core = mutate(a, rng.choice([0.0, 0.03, 0.1]), rng)
            if hfree:
                flank = lambda: bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 40)))  # noqa: E731
                b = flank() + core + flank()
            else:
                b = core
            n = len(b)
            W = rng.randint(0, 24)
            dmin, dmax = -W - max(0, m - n), W + max(0, n - m)
            if K + dmax - dmin > 15 * (K + 1):
                continue
            rc = rng.random() < 0.3
            view = b[::-1].translate(COMP) if rc else b
            pairs.append((a, b, dmin, dmax, rc))
            wants.append(orc.gotoh_str(a, view, hfree, 0, SC) + (W, m, n))
        if not pairs:
            continue
        got, err = emu.run_band16(pairs, SC, hfree, K, 0, True)
        for (gs, gb, _), (ws, wb, W, m, n) in zip(got, wants):
            checked += 1
            assert gs <= ws  # a band can only lose paths
            # a path that leaves the band makes more than W gap steps (twice that, and a second gap open, when both ends are fixed)
            bound = SC[0] * m + SC[3] * (W + 1) * (1 if hfree else 2) + (0 if hfree else SC[2])
            if gs > bound:
                certified += 1
                assert (gs, gb) == (ws, wb), (K, hfree, m, n, W)
    assert checked > 150 and certified > 60


def test_band16_traceback_profile_rows():
    """profile rows through the table (the final alignments of `tracy align`, sage.h:260): gotoh(profile, _createProfile(slice))"""
    from tracy_amd import hostlib
    refs, profs, rev = hostlib.synth_align(99, 6, 400, 260, 0)
    for K in (4, 8, 12):
        pairs, wants = [], []
        for i in range(4):
            p = np.ascontiguousarray(profs[i][:, :200 + 13 * i])
            ref = refs[i].tobytes()
            view = ref[::-1].translate(COMP) if rev[i] else ref
            ws, wb = orc.gotoh_prof(p, orc.create_profile_str(view), 1, 0, SC)
            lead, ce = ends_of(wb, len(view))
            m = p.shape[1]
            lo = max(0, lead - 20)
            sl = view[lo:min(len(view), ce + 20)]  # a slice around the aligned region, as trimReferenceSlice cuts it
            ws, wb = orc.gotoh_prof(p, orc.create_profile_str(sl), 1, 0, SC)
            W = 12
            raw = sl[::-1].translate(COMP) if rev[i] else sl  # what the kernel reads: the stored strand, viewed through the flag
            pairs.append((p, raw, -W - max(0, m - len(sl)), W + max(0, len(sl) - m), bool(rev[i])))
            wants.append((ws, wb))
        if any(K + p_[3] - p_[2] > 15 * (K + 1) for p_ in pairs):  # the band kernels' domain: a strip is done before the lane's next one is due
            continue
        got, err = emu.run_band16(pairs, SC, 1, K, 0, False)
        assert err == 0
        for (gs, gb, _), (ws, wb) in zip(got, wants):
            assert (gs, gb) == (ws, wb), K


def test_band16_origin_sweep_on_the_a_priori_band():
    """the origin-tracking sweep (KIND 1) on the band a known score allows: an alignment that ends in column c_e of row m with score
    S* has at most g = (match m - S*) / |ge| gap steps, so it lies on the diagonals c_e - m - g .. c_e - m + g; score and both ends
    equal those of the whole window (trimReferenceSlice, fmindex.h:429-463)"""
    rng = random.Random(5)
    done = 0
    for it in range(60):
        K = rng.choice([4, 8])
        pairs, wants = [], []
        for q in range(rng.randint(1, 4)):
            m = rng.randint(5, 180)
            a = bytes(rng.choice(b"ACGT") for _ in range(m))
            flank = lambda k: bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, k)))  # noqa: E731
            b = flank(150) + mutate(a, rng.choice([0.0, 0.04]), rng) + flank(150)
            rc = rng.random() < 0.3
            view = b[::-1].translate(COMP) if rc else b
            ws, wb = orc.gotoh_str(a, view, 1, 0, SC)
            lead, ce = ends_of(wb, len(view))
            g = gap_budget(ws, m, 1)
            dmin, dmax = ce - m - g - 1, ce - m + g + 1
            if ce == 0 or K + dmax - dmin > 15 * (K + 1):
                continue
            pairs.append((a, b, dmin, dmax, rc))
            wants.append((ws, lead, ce))
        if not pairs:
            continue
        got, err = emu.run_band16(pairs, SC, 1, K, 1, True)
        for (gs, _, ge), (ws, lead, ce) in zip(got, wants):
            assert (gs,) + ge == (ws, lead, ce)
            done += 1
    assert done > 60


def test_band16_band_without_the_path_scores_lower():
    """a band that does not hold the optimal path yields the best path inside it: a lower score, which the pipelines' certificate
    rejects (the pair is repeated on the whole matrix); the same pair on a band that holds the path equals the oracle"""
    rng = random.Random(2)
    x1 = bytes(rng.choice(b"ACGT") for _ in range(70))
    x2 = bytes(rng.choice(b"ACGT") for _ in range(80))
    ins = bytes(rng.choice(b"ACGT") for _ in range(40))
    a = x1 + ins + x2                                                # 40 rows the reference does not have: the path drops 40 diagonals
    b = x1 + x2 + bytes(rng.choice(b"ACGT") for _ in range(40))
    got, err = emu.run_band16([(a, b, -5, 5, False), (a, b, -48, 8, False)], SC, 1, 8, 0, True)
    ws, wb = orc.gotoh_str(a, b, 1, 0, SC)
    assert (got[1][0], got[1][1]) == (ws, wb)
    assert got[0][0] < ws and got[0][0] <= SC[0] * len(a) + SC[3] * 6


def test_band16_trailing_run_longer_than_the_lanes_period():
    """the last strip sweeps row m on to column n; with a long free trailing run its window outlasts the 16 (K + 1) steps after
    which a lane would turn to its next strip (K = 4: 80 steps)"""
    rng = random.Random(9)
    for K in (4, 8):
        for tail in (30, 90, 200):
            for g in (0, 24, 30):
                m = rng.randint(50, 230)
                a = bytes(rng.choice(b"ACGT") for _ in range(m))
                lead = rng.randint(0, 60)
                b = bytes(rng.choice(b"ACGT") for _ in range(lead)) + a + bytes(rng.choice(b"ACGT") for _ in range(tail))
                ws, wb = orc.gotoh_str(a, b, 1, 0, SC)
                l2, ce = ends_of(wb, len(b))
                d1 = ce - m
                got, err = emu.run_band16([(a, b, d1 - g - 1, d1 + g + 1, False)], SC, 1, K, 0, True)
                assert (got[0][0], got[0][1]) == (ws, wb), (K, tail, g, m)
                got, err = emu.run_band16([(a, b, d1 - g - 1, d1 + g + 1, False)], SC, 1, K, 1, True)
                assert (got[0][0], got[0][2]) == (ws, (l2, ce)), (K, tail, g, m, "origin")


def test_band16_quad_form_equals_the_row_form():
    """narrow bands (windows of at most three blocks of strip height 4: at most 12 diagonals) swept four lanes to a pair, sixteen pairs
    to a wave (band16_body P = 4): scores, traceback strings and the origin-tracking sweep's two ends are those of the sixteen-lane
    form on the same band -- and, where the band is certified, the whole matrix's (gotoh.h:71-175)"""
    rng = random.Random(44)
    compared = certified = 0
    for it in range(40):
        hfree = 1 if it % 4 else 0
        kind = it % 2
        pairs, wants = [], []
        for q in range(rng.randint(1, 16)):
            m = rng.randint(1, 90 if it % 3 else 12)
            a = bytes(rng.choice(b"ACGTN") if rng.random() < 0.03 else rng.choice(b"ACGT") for _ in range(m))
            b = mutate(a, rng.choice([0.0, 0.0, 0.02]), rng)
            if hfree and rng.random() < 0.5:
                b = b + bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 30)))  # columns right of the band: the last strip runs on along row m
            n = len(b)
            W = rng.randint(0, 5)
            d1 = n - m if not hfree else rng.choice([0, min(n - m, 2)])
            dmin, dmax = min(0, d1) - W, max(0, d1) + W
            if dmax - dmin + 1 > 12:
                continue
            rc = rng.random() < 0.3
            view = b[::-1].translate(COMP) if rc else b
            pairs.append((a, b, dmin, dmax, rc))
            wants.append(orc.gotoh_str(a, view, hfree, 0, SC) + (W, m, n))
        if not pairs:
            continue
        got, err = emu.run_band16(pairs, SC, hfree, 44, kind, True)
        ref = []
        for i in range(0, len(pairs), 4):
            r, e16 = emu.run_band16(pairs[i:i + 4], SC, hfree, 4, kind, True)
            ref += r
            assert (e16 != 0) == (err != 0) or len(pairs) > 4
        for (g, r, (ws, wb, W, m, n)) in zip(got, ref, wants):
            compared += 1
            assert g == r, (it, kind, hfree, m, n, W)
            bound = SC[0] * m + SC[3] * (W + 1) * (1 if hfree else 2) + (0 if hfree else SC[2])
            if kind == 0 and g[1] is not None and len(g[1]) and g[0] > bound and not hfree:
                certified += 1
                assert (g[0], g[1]) == (ws, wb)
    assert compared > 150 and certified > 10, (compared, certified)
