"""The pruned orientation sweep (tracy_amd/csrc/front.h) on the host emulator: the prefix sweep's kept row is row R of the matrix,
the band kernels continue from it, and a certified score / c_e is gotohScore's (sage.h:239-240) and the first column of row m that
reaches it -- on traces that match their window, on windows that hold the target twice, on noise."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

SC = (3, -5, -10, -4)
COMP = bytes.maketrans(b"ACGT", b"TGCA")
CODE = {65: 0, 67: 1, 71: 2, 84: 3, 78: 4}


def trace_profile(seq, rng, noise):
    m = len(seq)
    p = np.zeros((6, m), np.float32)
    for i, ch in enumerate(seq):
        w = np.array([rng.random() * noise for _ in range(4)], np.float32)
        w[CODE[ch]] += 1.0
        p[:4, i] = w / w.sum()
    return p


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
            out.append(rng.choice(b"ACGT"))
        else:
            out.append(ch)
    return bytes(out)


def matrix(q, codes, R, sc):
    """plain Gotoh, AlignConfig<true,false> (gotoh.h:71-175): row R as (H, F) lists, the score and the first column of row m reaching it"""
    go, ge = sc[2], sc[3]
    m, n = len(q), len(codes)
    NEG = -10 ** 9
    H = [0] * (n + 1)
    F = [NEG] * (n + 1)
    keep = None
    for r in range(1, m + 1):
        last = r == m
        Hn = [go + r * ge] + [0] * n
        Fn = [NEG] * (n + 1)
        E = NEG
        for c in range(1, n + 1):
            E = max(Hn[c - 1] + (0 if last else go + ge), E + (0 if last else ge))
            f = max(H[c] + go + ge, F[c] + ge)
            Hn[c] = max(H[c - 1] + int(q[r - 1][codes[c - 1]]), E, f)
            Fn[c] = f
        H, F = Hn, Fn
        if r == R:
            keep = (list(H), list(F))
    best = max(H[1:])
    return keep, best, 1 + H[1:].index(best)


def make_case(rng, kind, R, Kb):
    m = rng.randint(R + Kb + 12, R + 150)
    seq = bytes(rng.choice(b"ACGT") for _ in range(m))
    prof = trace_profile(seq, rng, rng.choice([0.05, 0.3]))
    flank = lambda k: bytes(rng.choice(b"ACGT") for _ in range(k))  # noqa: E731
    if kind == "match":
        ref = flank(rng.randint(0, 300)) + mutate(seq, rng.choice([0.0, 0.03, 0.08]), rng) + flank(rng.randint(0, 300))
    elif kind == "twice":  # the target twice, the better copy second or first
        a, b = mutate(seq, 0.06, rng), mutate(seq, 0.02, rng)
        if rng.random() < 0.5:
            a, b = b, a
        ref = flank(rng.randint(0, 60)) + a + flank(rng.randint(20, 200)) + b + flank(rng.randint(0, 60))
    elif kind == "indel":  # a long insertion / deletion below the prefix rows: the path leaves any narrow band
        cut = rng.randint(R + 5, m - 5)
        core = seq[:cut] + (flank(rng.randint(5, 60)) if rng.random() < 0.5 else b"") + seq[cut + (rng.randint(5, 40) if rng.random() < 0.5 else 0):]
        ref = flank(rng.randint(0, 200)) + core + flank(rng.randint(0, 200))
    else:  # noise
        ref = flank(rng.randint(m // 2, m + 300))
    if rng.random() < 0.15:
        ref = bytearray(ref)
        ref[rng.randrange(len(ref))] = ord("N")
        ref = bytes(ref)
    rc = rng.random() < 0.4
    given = ref.translate(COMP)[::-1] if rc else ref  # the kernels read a reverse-complement view of what they are given
    return prof, given, rc, ref


def test_front_rows_scores_and_certificates():
    rng = random.Random(5)
    certified = {"match": 0, "twice": 0, "indel": 0, "noise": 0}
    total = dict(certified)
    for it in range(36):
        Kp, Kb = rng.choice([(4, 4), (4, 8), (4, 12)])
        GLp = rng.choice([8, 16])  # the two prefix shapes: 8 x K (strand bounds) and 16 x 8 in the library (front.h)
        R = GLp * Kp
        halfw = rng.choice([8, 20, (15 * (Kb + 1) - Kb) // 2 - 1])
        kinds = [rng.choice(["match", "match", "twice", "indel", "noise"]) for _ in range(rng.randint(1, 4))]
        cases = [make_case(rng, k, R, Kb) for k in kinds]
        res, err = emu.run_front([c[0] for c in cases], [c[1] for c in cases], SC, Kp, Kb, halfw, revcomp=[c[2] for c in cases], want_rows=True, GLp=GLp)
        assert err == 0
        for kind, (prof, given, rc, ref), r in zip(kinds, cases, res):
            q = emu.table_rows(prof, SC)
            codes = [CODE.get(ch, 5) for ch in ref]
            (Hr, Fr), best, ce = matrix(q, codes, R, SC)
            n = len(ref)
            goe = SC[2] + SC[3]
            for c in range(1, n + 1):  # the kept row is row R
                x = int(r["row"][c])
                h = ((x & 0xffff) ^ 0x8000) - 0x8000 - goe
                f = (x >> 16 ^ 0x8000) - 0x8000
                assert (h, f) == (Hr[c], Fr[c]), (it, kind, c)
            v = [max(Hr[c], Fr[c]) for c in range(1, n + 1)]
            assert r["vmax"] == max(v) and r["cstar"] == 1 + v.index(max(v))
            assert r["score"] <= best  # a band holds real paths only
            total[kind] += 1
            if r["ok"]:
                certified[kind] += 1
                assert (r["score"], r["c_e"]) == (best, ce), (it, kind, r, best, ce)
    assert certified["match"] > 0.7 * total["match"] and total["match"] > 15, (certified, total)
    assert certified["noise"] == 0, (certified, total)


def het_profile(seq, alt, rng):
    """a trace whose positions from a breakpoint on show two equal peaks (the bases of two alleles): such a row scores
    int(0.5 match + 0.5 mismatch) = -1 against either base with 3 / -5, below the 0 the first certificate allows it"""
    m = len(seq)
    p = np.zeros((6, m), np.float32)
    for i in range(m):
        a, b = CODE[seq[i]], CODE[alt[i]]
        if a == b:
            p[a, i] = 1.0
        else:
            p[a, i] = 0.5
            p[b, i] = 0.5
    return p


def test_second_certificate_serves_heterozygous_rows():
    """front_certify_body's second bound (rows allowed max(row maximum, -1), vertical steps losing |ge| - 1 against it): traces whose rows
    below the prefix are heterozygous certify with it and not without it -- and whatever certifies is the matrix's score and c_e"""
    rng = random.Random(23)
    with_second = without = checked = 0
    for it in range(10):
        Kp, Kb, GLp = 4, 4, 16  # (the narrowest strips: their widest band pays for 140 lost points, less than the slack of ~200 heterozygous rows)
        R = GLp * Kp
        halfw = (15 * (Kb + 1) - Kb) // 2 - 1
        cases = []
        for _ in range(rng.randint(2, 4)):
            m = rng.randint(R + 230, R + 320)
            seq = bytes(rng.choice(b"ACGT") for _ in range(m))
            bp = rng.randint(R // 2, R + 30)            # the second allele is the first one shifted by an indel from here on
            shift = rng.randint(1, 9)
            alt = seq[:bp] + seq[bp + shift:] + bytes(rng.choice(b"ACGT") for _ in range(shift))
            prof = het_profile(seq, alt, rng)
            flank = lambda k: bytes(rng.choice(b"ACGT") for _ in range(k))  # noqa: E731
            ref = flank(rng.randint(0, 250)) + mutate(seq, rng.choice([0.0, 0.02]), rng) + flank(rng.randint(0, 250))
            rc = rng.random() < 0.4
            cases.append((prof, ref.translate(COMP)[::-1] if rc else ref, rc, ref))
        for second in (True, False):
            res, err = emu.run_front([c[0] for c in cases], [c[1] for c in cases], SC, Kp, Kb, halfw, revcomp=[c[2] for c in cases], GLp=GLp,
                                     second_bound=second)
            assert err == 0
            for (prof, given, rc, ref), r in zip(cases, res):
                if not r["ok"]:
                    continue
                q = emu.table_rows(prof, SC)
                _, best, ce = matrix(q, [CODE.get(ch, 5) for ch in ref], R, SC)
                assert (r["score"], r["c_e"]) == (best, ce), (it, second, r, best, ce)
                checked += 1
                if second:
                    with_second += 1
                else:
                    without += 1
    assert with_second >= without + 10 and checked > 20, (with_second, without, checked)


def test_band_on_16_bit_cells_equals_the_tagged_form():
    """band16_cont16_body (the band below the kept row on v_add_u16 / v_max_i16 cells, values kept as H + go + ge) against band16_body
    CONT (tagged int32 recurrence): the same score and c_e for EVERY pair -- certified or not, both strands, every strip height, row m
    in any slot of the last strip -- and the matrix's where the certificate holds; several scorings"""
    rng = random.Random(99)
    same = certified = 0
    for it in range(30):
        Kp, Kb = rng.choice([(4, 4), (4, 8), (4, 12)])
        GLp = rng.choice([8, 16])
        R = GLp * Kp
        halfw = rng.choice([8, 20, (15 * (Kb + 1) - Kb) // 2 - 1])
        score = rng.choice([SC, SC, (5, -4, -10, -1), (2, -3, -5, -2), (1, -1, -2, -1)])
        kinds = [rng.choice(["match", "match", "twice", "indel", "noise"]) for _ in range(rng.randint(1, 4))]
        cases = [make_case(rng, k, R, Kb) for k in kinds]
        args = ([c[0] for c in cases], [c[1] for c in cases], score, Kp, Kb, halfw)
        wide, err0 = emu.run_front(*args, revcomp=[c[2] for c in cases], GLp=GLp)
        narrow, err1 = emu.run_front(*args, revcomp=[c[2] for c in cases], GLp=GLp, cont16=True)
        assert err0 == 0 and err1 == 0
        for kind, (prof, given, rc, ref), a, b in zip(kinds, cases, wide, narrow):
            if a["score"] <= -5000:  # no path of the band reaches row m: both forms show their -inf (and certify nothing)
                assert b["score"] <= -15000 and a["ok"] == b["ok"] == 0, (it, kind, a, b)
                continue
            assert (a["score"], a["c_e"], a["ok"], a["cstar"], a["shift"]) == (b["score"], b["c_e"], b["ok"], b["cstar"], b["shift"]), (it, kind, a, b)
            same += 1
            if b["ok"]:
                q = emu.table_rows(prof, score)
                _, best, ce = matrix(q, [CODE.get(ch, 5) for ch in ref], R, score)
                assert (b["score"], b["c_e"]) == (best, ce), (it, kind, b, best, ce)
                certified += 1
    assert same > 50 and certified > 15, (same, certified)


def test_quad_tier_of_the_pruned_sweep_equals_the_row_form():
    """the narrow first tier (c* +- 5 on strips of four rows, four lanes per pair: band16_cont16_body P = 4) against the sixteen-lane
    form on the same band: the same score, c_e and verdict for every pair, and the matrix's where the certificate holds"""
    rng = random.Random(404)
    same = certified = 0
    for it in range(30):
        Kp, Kb = 4, 4
        GLp = rng.choice([8, 16])
        R = GLp * Kp
        halfw = rng.choice([2, 4, 5])
        score = rng.choice([SC, SC, (5, -4, -10, -1), (2, -3, -5, -2)])
        kinds = [rng.choice(["match", "match", "match", "indel", "noise"]) for _ in range(rng.randint(1, 4))]
        cases = [make_case(rng, k, R, Kb) for k in kinds]
        args = ([c[0] for c in cases], [c[1] for c in cases], score, Kp, Kb, halfw)
        rows16, err0 = emu.run_front(*args, revcomp=[c[2] for c in cases], GLp=GLp, cont16=True)
        quads, err1 = emu.run_front(*args, revcomp=[c[2] for c in cases], GLp=GLp, cont16=True, quad=True)
        assert err0 == 0 and err1 == 0
        for kind, (prof, given, rc, ref), a, b in zip(kinds, cases, rows16, quads):
            if a["score"] <= -5000:
                assert b["score"] <= -5000 and a["ok"] == b["ok"] == 0, (it, kind, a, b)
                continue
            assert (a["score"], a["c_e"], a["ok"], a["cstar"], a["shift"]) == (b["score"], b["c_e"], b["ok"], b["cstar"], b["shift"]), (it, kind, a, b)
            same += 1
            if b["ok"]:
                q = emu.table_rows(prof, score)
                _, best, ce = matrix(q, [CODE.get(ch, 5) for ch in ref], R, score)
                assert (b["score"], b["c_e"]) == (best, ce), (it, kind, b, best, ce)
                certified += 1
    assert same > 40 and certified > 10, (same, certified)
