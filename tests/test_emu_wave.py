"""The wave-level kernel bodies (tracy_amd/csrc/dp_kernels.h) executed on the host by a 64-lane
lock-step emulator (one fiber per lane), compared bit for bit with the oracle.  This checks everything in the HIP kernels
except the __global__ wrappers, the DPP shift and the memory system."""
import os
import sys

import numpy as np
import pytest

import pyoracle as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

SC = (3, -5, -10, -4)
CONFIGS = [(0, 0), (1, 0), (0, 1), (1, 1)]


def rand_seq(rng, n, alpha=b"ACGT"):
    return bytes(rng.choice(list(alpha), size=n).tolist())


def rand_profile(rng, n, sharp=True):
    p = np.zeros((6, n), dtype=np.float32)
    x = rng.random((4, n)).astype(np.float32)
    if sharp:
        x = x ** 6
    p[:4] = x / x.sum(axis=0, keepdims=True)
    return p


def mutate(rng, s, rate=0.1):
    out = bytearray()
    for ch in s:
        u = rng.random()
        if u < rate / 3:
            continue
        if u < 2 * rate / 3:
            out.append(int(rng.choice(list(b"ACGT"))))
        if u < rate:
            out.append(int(rng.choice(list(b"ACGT"))))
        else:
            out.append(ch)
    return bytes(out)


@pytest.mark.parametrize("K", [4, 8, 16])
@pytest.mark.parametrize("cfg", CONFIGS)
def test_char_mode(K, cfg):
    rng = np.random.default_rng(100 + K + cfg[0] * 2 + cfg[1])
    sizes = [(0, 5), (5, 0), (1, 1), (3, 70), (70, 3), (64 * K // 8, 40), (50, 90)]
    if K == 4:
        sizes += [(300, 120), (257, 33)]  # more than one pass (64*K = 256 rows)
    for (m, n) in sizes:
        alpha = b"AC" if (m + n) % 3 == 0 else b"ACGT"
        s2 = rand_seq(rng, n, alpha)
        s1 = mutate(rng, s2[: max(m, 0)] + rand_seq(rng, max(0, m - n), alpha))[:m] if m else b""
        s1 = (s1 + rand_seq(rng, m, alpha))[:m]
        want = orc.gotoh_str(s1, s2, cfg[0], cfg[1], SC)
        got = emu.run(s1, s2, SC, cfg[0], cfg[1], emu.MODE_CHAR, K, trace=True)
        assert (got[0], got[1]) == want, (m, n, K, cfg)
        got_s = emu.run(s1, s2, SC, cfg[0], cfg[1], emu.MODE_CHAR, K, trace=False)
        assert got_s[0] == want[0]


@pytest.mark.parametrize("K", [4, 16])
def test_qp_mode_and_revcomp(K):
    rng = np.random.default_rng(7 + K)
    for (m, n) in [(1, 9), (40, 120), (64, 64), (130, 50)] + ([(300, 80)] if K == 4 else []):
        p1 = rand_profile(rng, m)
        ref = rand_seq(rng, n, b"ACGTACGTACGTNn-x")
        p2 = orc.create_profile_str(ref)
        for cfg in [(1, 0), (1, 1), (0, 0)]:
            want = orc.gotoh_prof(p1, p2, cfg[0], cfg[1], SC)
            got = emu.run(p1, ref, SC, cfg[0], cfg[1], emu.MODE_QP, K, trace=True)
            assert (got[0], got[1]) == want and got[2] == 0
            assert emu.run(p1, ref, SC, cfg[0], cfg[1], emu.MODE_QP, K, trace=False)[0] == want[0]
        # reverse-complemented reference (sage.h:236-240): read backwards + complemented codes
        want = orc.gotoh_score_prof(p1, orc.revcomp_profile(p2), 1, 0, SC)
        assert emu.run(p1, ref, SC, 1, 0, emu.MODE_QP, K, trace=False, revcomp=True)[0] == want
    # trimmed view of a full profile (row stride stays the full length)
    p1 = rand_profile(rng, 60)
    ref = rand_seq(rng, 90)
    want = orc.gotoh_prof(p1[:, 7:52], orc.create_profile_str(ref), 1, 0, SC)
    got = emu.run(p1, ref, SC, 1, 0, emu.MODE_QP, K, trace=True, a1_view=(7, 45))
    assert (got[0], got[1]) == want


@pytest.mark.parametrize("K", [4, 8])
def test_profile_profile_mode(K):
    rng = np.random.default_rng(21 + K)
    for (m, n) in [(5, 7), (33, 70), (100, 20)] + ([(270, 30)] if K == 4 else []):
        p1, p2 = rand_profile(rng, m, sharp=False), rand_profile(rng, n)
        for cfg in [(1, 0), (1, 1)]:
            want = orc.gotoh_prof(p1, p2, cfg[0], cfg[1], SC)
            got = emu.run(p1, p2, SC, cfg[0], cfg[1], emu.MODE_PROF, K, trace=True)
            assert (got[0], got[1]) == want
            assert emu.run(p1, p2, SC, cfg[0], cfg[1], emu.MODE_PROF, K, trace=False)[0] == want[0]
            # the score kernel with 16-bit cells (any AlignConfig, multi-pass strips), with and without the screened score
            assert emu.run(p1, p2, SC, cfg[0], cfg[1], emu.MODE_PROF, K, trace=False, narrow=True)[0] == want[0]
            assert emu.run(p1, p2, SC, cfg[0], cfg[1], emu.MODE_PROF, K, trace=False, narrow=True, screen=True)[0] == want[0]
        for cfg in [(0, 0), (0, 1)]:
            assert emu.run(p1, p2, SC, cfg[0], cfg[1], emu.MODE_PROF, K, trace=False, narrow=True)[0] == orc.gotoh_prof(p1, p2, cfg[0], cfg[1], SC)[0]
    # alignment profiles carry weight in rows 4 ('N') and 5 ('-'): the 25-term path (row 4 is not all zero)
    p1, p2 = rand_profile(rng, 40, sharp=False), rand_profile(rng, 55)
    p1[4, 7] = np.float32(0.25)
    p1[5, 9] = np.float32(0.5)
    for q in (p1, p2):
        for cfg in [(1, 1)]:
            want = orc.gotoh_prof(q, p2, cfg[0], cfg[1], SC)
            got = emu.run(q, p2, SC, cfg[0], cfg[1], emu.MODE_PROF, K, trace=True)
            assert (got[0], got[1]) == want
    p2[4, 54] = np.float32(1.0)  # only the second profile has an N column
    want = orc.gotoh_prof(rand_profile(np.random.default_rng(1), 30), p2, 1, 1, SC)
    got = emu.run(rand_profile(np.random.default_rng(1), 30), p2, SC, 1, 1, emu.MODE_PROF, K, trace=True)
    assert (got[0], got[1]) == want


def screen_columns(rng, n, kind, nt):
    """profile columns [n][5] (rows A C G T N) of mass <= 1.001"""
    c = np.zeros((n, 5), dtype=np.float32)
    if kind == "trace":  # createProfile (profile.h:22-52): called bases share normfac, the rest gets (1 - normfac) / 4
        sig = rng.random((n, 4)).astype(np.float32) ** 8
        sig[np.arange(n), rng.integers(0, 4, n)] += 1.0
        called = sig >= 0.33 * sig.max(axis=1, keepdims=True)
        tot = (sig * called).sum(axis=1, keepdims=True)
        normfac = (tot / sig.sum(axis=1, keepdims=True)).astype(np.float32)
        c[:, :4] = normfac * (sig * called / tot) + (np.float32(1) - normfac) * np.float32(0.25)
    elif kind == "dyadic":  # sixteenths: products and sums are exact, scores sit on or near integers
        k = rng.multinomial(16, [0.25] * nt, size=n)
        c[:, :nt] = k / 16.0
    elif kind == "onehot":
        c[np.arange(n), rng.integers(0, nt, n)] = 1.0
    elif kind == "uniform":
        c[:, :4] = 0.25
    elif kind == "heavy":  # up to the admitted mass
        x = rng.random((n, nt)).astype(np.float32) ** 3
        c[:, :nt] = x / x.sum(axis=1, keepdims=True) * np.float32(1.0009)
    else:  # consensus-like: averages over a few sequences, weight on N for nt = 5
        x = rng.integers(0, 6, (n, nt)).astype(np.float32)
        x[:, 0] += 1
        c[:, :nt] = x / x.sum(axis=1, keepdims=True)
    return c


def test_screened_profile_score_never_differs():
    """SubProf::screen: whenever the short form claims trunc(score), it IS the int of the 25-term float chain (align.h:112-117)"""
    rng = np.random.default_rng(2718)
    n = 25000
    for nt in (4, 5):
        kinds = ["trace", "dyadic", "onehot", "uniform", "heavy", "consensus"]
        for ka in kinds:
            for kb in ("trace", "dyadic", "onehot", "heavy"):
                a, b = screen_columns(rng, n, ka, nt), screen_columns(rng, n, kb, nt)
                for sc in ((3, -5), (5, -4), (1, -1), (1000, -1000), (2, 3), (0, 0), (-7, 11)):
                    unproven, wrong = emu.screen_check(a, b, sc[0], sc[1], nt)
                    assert wrong == 0, (nt, ka, kb, sc)
                    if ka == "trace" and kb == "trace" and sc in ((3, -5), (5, -4), (1, -1)):
                        assert unproven < n * 2e-3, (nt, sc, unproven)  # realistic columns almost never need the chain


def test_profile_profile_screened_sweep():
    """the sweep with screening on: same alignments -- screened strips, tabulated one-hot / uniform columns, slots that fall
    back to the float chain, and the give-up switch (a profile of columns whose scores sit on integers)"""
    rng = np.random.default_rng(99)

    def mixed(k, kinds, nt=4):
        cols = np.concatenate([screen_columns(rng, k, kd, nt) for kd in kinds])
        cols = cols[rng.permutation(len(cols))[:k]]
        return np.ascontiguousarray(np.vstack([cols.T, np.zeros((1, k), np.float32)]))

    cases = [(mixed(33, ["trace"]), mixed(70, ["trace"])),
             (mixed(100, ["trace", "onehot", "uniform"]), mixed(90, ["trace", "onehot", "uniform", "dyadic"])),
             (mixed(70, ["onehot"]), mixed(130, ["onehot", "uniform"])),
             (mixed(60, ["dyadic"]), mixed(200, ["dyadic"])),
             (mixed(40, ["consensus", "onehot"], 5), mixed(75, ["consensus", "onehot", "trace"], 5)),
             (mixed(600, ["trace", "onehot"]), mixed(90, ["trace", "uniform"]))]  # two passes at K = 8
    for p1, p2 in cases:
        for cfg in [(1, 0), (1, 1)]:
            want = orc.gotoh_prof(p1, p2, cfg[0], cfg[1], SC)
            got = emu.run(p1, p2, SC, cfg[0], cfg[1], emu.MODE_PROF, 8, trace=True, screen=True)
            assert (got[0], got[1]) == want
            assert emu.run(p1, p2, SC, cfg[0], cfg[1], emu.MODE_PROF, 8, trace=False, screen=True)[0] == want[0]
    p1, p2 = cases[1]
    assert emu.run(p1, p2, SC, 1, 1, emu.MODE_PROF, 4, trace=False, screen=True)[0] == orc.gotoh_prof(p1, p2, 1, 1, SC)[0]


@pytest.mark.parametrize("K", [4, 16])
def test_needle(K):
    sc = (5, -4, -10, -1)
    rng = np.random.default_rng(31 + K)
    for (m, n) in [(0, 3), (4, 0), (9, 9), (40, 77), (90, 30)] + ([(260, 40)] if K == 4 else []):
        s1, s2 = rand_seq(rng, m, b"AC"), rand_seq(rng, n, b"AC")
        for cfg in CONFIGS:
            want = orc.needle_str(s1, s2, cfg[0], cfg[1], sc)
            got = emu.run(s1, s2, sc, cfg[0], cfg[1], emu.MODE_CHAR, K, trace=True, needle=True)
            assert (got[0], got[1]) == want
            assert emu.run(s1, s2, sc, cfg[0], cfg[1], emu.MODE_CHAR, K, trace=False, needle=True)[0] == want[0]
    p1, p2 = rand_profile(rng, 20, sharp=False), rand_profile(rng, 33, sharp=False)
    want = orc.needle_prof(p1, p2, 1, 1, sc)
    got = emu.run(p1, p2, sc, 1, 1, emu.MODE_PROF, K, trace=True, needle=True)
    assert (got[0], got[1]) == want


def test_qp_overflow_flag():
    rng = np.random.default_rng(5)
    p1 = rand_profile(rng, 10)
    _, _, err = emu.run(p1, b"ACGTACGT", (3000, -5, -10, -4), 1, 0, emu.MODE_QP, 4, trace=True)
    assert err & 1


@pytest.mark.parametrize("K", [4, 16])
def test_narrow_score_kernel(K):
    """the 16-bit score-only formulation gives the same scores.  Its domain (checked by narrow_ok in the
    C ABI): AlignConfig<true,false>, ge < 0, go <= 0, one pass of the strip height."""
    rng = np.random.default_rng(77 + K)
    for (m, n) in [(1, 40), (2, 7), (63, 300), (130 if K == 16 else 200, 90), (64 * K, 50), (64 * K - 5, 77)]:
        p1 = rand_profile(rng, m)
        if m > 2:  # weight in row 4 ('N'): the entries of N columns are not a constant of the scoring then
            j = rng.integers(0, m, size=max(1, m // 7))
            p1[4, j] = p1[0, j]
            p1[0, j] = 0
        ref = rand_seq(rng, n, b"ACGTACGTACGTNn-x")
        p2 = orc.create_profile_str(ref)
        want = orc.gotoh_score_prof(p1, p2, 1, 0, SC)
        assert emu.run(p1, ref, SC, 1, 0, emu.MODE_QP, K, trace=False, narrow=True)[0] == want, (m, n)
        want = orc.gotoh_score_prof(p1, orc.revcomp_profile(p2), 1, 0, SC)
        assert emu.run(p1, ref, SC, 1, 0, emu.MODE_QP, K, trace=False, revcomp=True, narrow=True)[0] == want
        s1 = rand_seq(rng, m, b"ACGT")
        s2 = rand_seq(rng, n, b"ACGT")
        assert emu.run(s1, s2, SC, 1, 0, emu.MODE_CHAR, K, trace=False, narrow=True)[0] == orc.gotoh_score_str(s1, s2, 1, 0, SC)
    # all-mismatch / long-gap extremes stay inside the int16 window
    s1, s2 = b"A" * 250, b"C" * 400
    assert emu.run(s1, s2, SC, 1, 0, emu.MODE_CHAR, K, trace=False, narrow=True)[0] == orc.gotoh_score_str(s1, s2, 1, 0, SC)
    sc2 = (5, -4, -10, -1)
    assert emu.run(s1, s1[:100] + s2, sc2, 1, 0, emu.MODE_CHAR, K, trace=False, narrow=True)[0] == orc.gotoh_score_str(s1, s1[:100] + s2, 1, 0, sc2)


@pytest.mark.parametrize("K", [15])
def test_sweep16_compact_and_full_forms(K):
    """the 16-bit query-profile sweep exists with a four-code table (references of A C G T) and a six-code one; the block map
    of the encoders decides per pair.  Same scores / alignments from both, specials at either end of a 256-byte block"""
    rng = np.random.default_rng(500 + K)
    m, n = 12 * K, 520
    p1 = rand_profile(rng, m)
    base = rand_seq(rng, n, b"ACGT")
    q = orc.create_profile_str(mutate(rng, base[100:100 + m], 0.05)[:m].ljust(m, b"A"))
    for ref in (base, base[:255] + b"N" + base[256:], base[:256] + b"-" + base[257:], base[:n - 1] + b"n"):
        for prof in ((p1, q) if ref is base else (q,)):
            p2 = orc.create_profile_str(ref)
            for rc in (False, True):
                want = orc.gotoh_prof(prof, orc.revcomp_profile(p2) if rc else p2, 1, 0, SC)
                assert emu.run(prof, ref, SC, 1, 0, emu.MODE_QP, K, trace=False, narrow=True, revcomp=rc)[0] == want[0]
                got = emu.run_band(prof, ref, SC, 1, 0, emu.MODE_QP, K, B=64, narrow=True, revcomp=rc)
                assert (got[0], got[1]) == want


@pytest.mark.parametrize("K", [8, 15])
def test_sweep16_strings_through_the_table(K):
    """MODE_CQ in the 16-bit sweep: gotohScore(string, string) <true,false> with the rows in the table, both forms, reverse
    complement view, columns outside A C G T N (lower case included: byte equality)"""
    rng = np.random.default_rng(60 + K)
    for (m, n) in [(1, 30), (40, 200), (64 * K, 90), (17 * K + 3, 400)]:
        q = rand_seq(rng, m, b"ACGTACGTN")
        for alpha in (b"ACGT", b"ACGTACGTN", b"ACGTacgtNx-"):
            ref = rand_seq(rng, n, alpha)
            if m > 30 and n > m:
                ref = ref[:n // 3] + bytes(c if rng.random() > 0.1 else int(rng.choice(list(b"ACGT"))) for c in q[:min(m, n - n // 3)]) + ref[n // 3 + min(m, n - n // 3):]
                ref = ref[:n]
            for rc in (False, True):
                # reverseComplement (fmindex.h:8-24) rewrites A C G T only; every other byte keeps its value
                oriented = bytes({65: 84, 67: 71, 71: 67, 84: 65}.get(c, c) for c in reversed(ref)) if rc else ref
                want = orc.gotoh_score_str(q, oriented, 1, 0, SC)
                assert emu.run(q, ref, SC, 1, 0, emu.MODE_CQ, K, trace=False, narrow=True, revcomp=rc)[0] == want, (m, n, alpha, rc)


@pytest.mark.parametrize("K", [12, 15])
def test_odd_strip_heights(K):
    rng = np.random.default_rng(K)
    for (m, n) in [(1, 30), (64 * K, 70), (64 * K - 7, 90), (64 * K + 20, 40)]:
        p1 = rand_profile(rng, m)
        ref = rand_seq(rng, n, b"ACGTACGTN-")
        p2 = orc.create_profile_str(ref)
        want = orc.gotoh_prof(p1, p2, 1, 0, SC)
        got = emu.run(p1, ref, SC, 1, 0, emu.MODE_QP, K, trace=True)
        assert (got[0], got[1]) == want
        assert emu.run(p1, ref, SC, 1, 0, emu.MODE_QP, K, trace=False)[0] == want[0]
        if m <= 64 * K:
            assert emu.run(p1, ref, SC, 1, 0, emu.MODE_QP, K, trace=False, narrow=True)[0] == want[0]
        s1 = rand_seq(rng, m)
        want = orc.gotoh_str(s1, ref, 1, 1, SC)
        got = emu.run(s1, ref, SC, 1, 1, emu.MODE_CHAR, K, trace=True)
        assert (got[0], got[1]) == want


@pytest.mark.parametrize("K", [4, 15])
def test_band_traceback(K):
    """checkpointed score pass + band traceback == full-matrix traceback (same btr, same score)"""
    rng = np.random.default_rng(900 + K)
    cases = [(1, 1), (1, 50), (40, 1), (30, 200), (64 * K - 9, 150), (100, 97), (K + 1, 40), (2 * K - 1, 33)]
    if K == 4:
        cases.append((64 * K, 130))
    for (m, n) in cases:
        if m > 64 * K:
            continue
        ref = rand_seq(rng, n, b"ACGTACGTACGTN")
        start = int(rng.integers(0, max(1, n - m)))
        q = mutate(rng, ref[start:start + m] + rand_seq(rng, m), 0.08)[:m]
        q = (q + rand_seq(rng, m))[:m]
        idx = {65: 0, 67: 1, 71: 2, 84: 3}
        p1 = np.zeros((6, m), dtype=np.float32)
        for jx, ch in enumerate(q):
            col = rng.random(4).astype(np.float32) * np.float32(0.1)
            col[idx.get(ch, 0)] += np.float32(1.0)
            p1[:4, jx] = col / col.sum()
        p2 = orc.create_profile_str(ref)
        for B in ((16, 64) if m < 200 else (64,)):
            # band mode domain: free end gaps on the first/last row only (AlignConfig<true,false>), ge < 0
            for narrow in (True, False):
                want = orc.gotoh_prof(p1, p2, 1, 0, SC)
                got = emu.run_band(p1, ref, SC, 1, 0, emu.MODE_QP, K, B=B, narrow=narrow)
                assert (got[0], got[1]) == want and got[2] == 0, (m, n, K, B, narrow)
            want = orc.gotoh_str(q, ref, 1, 0, SC)
            got = emu.run_band(q, ref, SC, 1, 0, emu.MODE_CHAR, K, B=B, narrow=True)
            assert (got[0], got[1]) == want and got[2] == 0
        want = orc.gotoh_prof(p1, orc.revcomp_profile(p2), 1, 0, SC)
        got = emu.run_band(p1, ref, SC, 1, 0, emu.MODE_QP, K, B=32, narrow=True, revcomp=True)
        assert (got[0], got[1]) == want


@pytest.mark.parametrize("K", [4, 15])
def test_prefix_bound_kernel(K):
    """gotoh_prefix_body: eight pairs per wave, rows 1..8K; result == max over the columns of H and F of row 8K as
    computed by the row-state oracle (forward and reverse-complemented references, ragged lengths, partial waves)"""
    rng = np.random.default_rng(4000 + K)
    R = 8 * K
    for npairs in (8, 3):
        profs, refs, rc = [], [], []
        for i in range(npairs):
            m = R + 1 + int(rng.integers(0, 200))
            n = int(rng.integers(5, 400))
            profs.append(rand_profile(rng, m))
            refs.append(rand_seq(rng, n, b"ACGTACGTACGTN"))
            rc.append(bool(i % 2))
        # one pair whose trace really lies on the reference: the bound must reach high values there
        refs[0] = rand_seq(rng, 300, b"ACGT")
        idx = {65: 0, 67: 1, 71: 2, 84: 3}
        p = np.zeros((6, profs[0].shape[1]), np.float32)
        src = (refs[0] * 3)[20:20 + p.shape[1]]
        for j, ch in enumerate(src):
            col = rng.random(4).astype(np.float32) * np.float32(0.1)
            col[idx[ch]] += np.float32(1)
            p[:4, j] = col / col.sum()
        profs[0], rc[0] = p, False
        got, err = emu.run_prefix(profs, refs, SC, K, rc)
        assert err == 0
        for i in range(npairs):
            p2 = orc.create_profile_str(refs[i])
            if rc[i]:
                p2 = orc.revcomp_profile(p2)
            H, F = orc.gotoh_row_state(profs[i], p2, R, SC)
            assert got[i] == int(max(H.max(), F.max())), (K, npairs, i)
        assert got[0] > 0


def test_origin_sweep_ends():
    """gotoh_origin_body (the sweep behind trimReferenceSlice in the decompose pipeline): H(m,n), the number of leading
    'h' columns and the last column before the trailing 'h' run equal those of the reference's traceback string"""
    import emu
    import pyoracle as orc
    from sage_oracle import revcomp
    rng = np.random.default_rng(77)
    sc = (3, -5, -10, -4)

    def ends_of(btr, n):  # push-order string: trailing run first
        fwd = btr[::-1]
        lead = len(fwd) - len(fwd.lstrip(b"h")) if isinstance(fwd, bytes) else len(fwd) - len(fwd.lstrip("h"))
        trail = len(fwd) - len(fwd.rstrip(b"h")) if isinstance(fwd, bytes) else len(fwd) - len(fwd.rstrip("h"))
        return lead, n - trail
    cases = [(1, 1, 4), (3, 40, 4), (17, 9, 4), (60, 200, 4), (64, 130, 8), (200, 500, 8), (255, 300, 4), (300, 90, 8), (500, 600, 8)]
    for (m, n, K) in cases:
        for rep in range(3):
            ref = bytes(rng.choice(list(b"ACGT" if rep else b"AC"), size=n).tolist())
            start = int(rng.integers(0, max(1, n - m + 1)))
            q = bytearray((ref[start:start + m] + bytes(rng.choice(list(b"ACGT"), size=m).tolist()))[:m])
            for j in range(m):  # substitutions and a little structure that creates ties (repeats, N)
                if rng.random() < 0.08:
                    q[j] = int(rng.choice(list(b"ACGTN")))
            q = bytes(q)
            rc = bool(rep == 2)
            oriented = revcomp(ref) if rc else ref
            want_score, want_btr = orc.gotoh_str(q, oriented, 1, 0, sc)
            got = emu.run_origin(q, ref, sc, K, revcomp=rc)
            assert got == (want_score,) + ends_of(want_btr, n), (m, n, K, rep, got, want_score, ends_of(want_btr, n))
            # table form (MODE_CQ: rows over ACGTN, columns as case-sensitive codes); a column letter no row can hold mismatches
            ref2 = ref if rep != 1 else (ref[:n // 3] + b"x" + ref[n // 3 + 1:n // 2] + b"a" + ref[n // 2 + 1:])[:n]
            o2 = (revcomp(ref2) if rc else ref2) if rep != 1 else ref2
            w2 = orc.gotoh_str(q, o2, 1, 0, sc)
            got = emu.run_origin(q, ref2, sc, K, revcomp=rc, table=True)
            assert got == (w2[0],) + ends_of(w2[1], n), (m, n, K, rep, "table")


def test_origin_sweep_with_profile_rows():
    """the origin-tracking sweep with a trace profile as a1 (MODE_QP table): score and the two ends of gotoh(profile,
    _createProfile(window)) <true,false>, as trimReferenceSlice reads them off the traceback (tracy align, sage.h:258-259)"""
    import emu
    import pyoracle as orc
    rng = np.random.default_rng(404)

    def ends_of(btr, n):
        fwd = btr[::-1]
        return len(fwd) - len(fwd.lstrip(b"h")), n - (len(fwd) - len(fwd.rstrip(b"h")))
    for (m, n, K) in [(1, 9, 4), (30, 200, 4), (64, 300, 8), (200, 700, 15), (15 * 64, 1200, 15), (500, 520, 8)]:
        for rep in range(3):
            ref = rand_seq(rng, n, b"ACGT" if rep != 1 else b"ACGTACGTN-x")
            start = int(rng.integers(0, max(1, n - m + 1)))
            src = (ref[start:start + m] + rand_seq(rng, m))[:m]
            p1 = np.zeros((6, m), np.float32)
            for j, ch in enumerate(src):
                col = rng.random(4).astype(np.float32) * np.float32(0.15)
                if ch in b"ACGT" and rng.random() > 0.05:
                    col[b"ACGT".index(ch)] += np.float32(1)
                p1[:4, j] = col / col.sum()
            rc = rep == 2
            p2 = orc.create_profile_str(ref)
            want = orc.gotoh_prof(p1, orc.revcomp_profile(p2) if rc else p2, 1, 0, SC)
            got = emu.run_origin_qp(p1, ref, SC, K, revcomp=rc)
            assert got == (want[0],) + ends_of(want[1], n), (m, n, K, rep)


def test_string_traceback_through_the_table():
    """MODE_CQ: gotoh(string, string) with the row chars in the query-profile table == the byte-compare kernel == the oracle,
    all AlignConfigs, reverse-complement view, columns with letters outside ACGTN (lower case included: byte equality)"""
    import emu
    import pyoracle as orc
    from sage_oracle import revcomp
    rng = np.random.default_rng(5)
    sc = (3, -5, -10, -4)
    for (m, n, K) in [(1, 1, 4), (30, 50, 4), (200, 260, 4), (255, 200, 8), (300, 400, 8), (500, 380, 8)]:
        ref = bytearray(rng.choice(list(b"ACGTN"), size=n).tolist())
        q = bytearray((bytes(ref)[:m] + bytes(rng.choice(list(b"ACGT"), size=m).tolist()))[:m])
        for j in range(0, m, 7):
            q[j] = int(rng.choice(list(b"ACGTN")))
        for j in range(3, n, 41):
            ref[j] = int(rng.choice(list(b"acgtRY-x")))
        q, ref = bytes(q), bytes(ref)
        for cfg in [(1, 0), (0, 0), (1, 1), (0, 1)]:
            want = orc.gotoh_str(q, ref, cfg[0], cfg[1], sc)
            got = emu.run(q, ref, sc, cfg[0], cfg[1], emu.MODE_CQ, K)
            assert (got[0], got[1]) == want and got[2] == 0, (m, n, K, cfg)
        clean = bytes(c if c in b"ACGTN" else ord("N") for c in ref)  # the reverse complement is defined on ACGTN
        want = orc.gotoh_str(q, revcomp(clean), 1, 0, sc)
        got = emu.run(q, clean, sc, 1, 0, emu.MODE_CQ, K, revcomp=True)
        assert (got[0], got[1]) == want


def test_banded_traceback_equals_the_whole_matrix_when_it_certifies():
    """PAIR_BANDED (final alignments of `tracy align`): strips of four rows, every pass sweeps the columns its rows reach inside
    the diagonal band.  Where the score beats top - |ge| (W + 1) (top = sum of the row maxima) the result is the whole
    matrix's: score and traceback equal the oracle's.  Slices longer and shorter than the trace, reverse-complement view,
    several passes; a band too narrow to certify must at least never score above the optimum"""
    rng = np.random.default_rng(77)
    K = 4
    certified = 0
    for (m, n, W, rate) in [(300, 320, 40, 0.006), (520, 500, 50, 0.006), (700, 760, 60, 0.006), (600, 600, 40, 0.004), (770, 780, 6, 0.05),
                            (257, 300, 30, 0.008), (513, 505, 40, 0.006)]:
        ref = rand_seq(rng, n, b"ACGT")
        src = ref if (m + n) % 2 else bytes({65: 84, 67: 71, 71: 67, 84: 65}[c] for c in reversed(ref))  # the trace reads one strand or the other
        q = mutate(rng, (src + rand_seq(rng, m))[:m + 40], rate)[:m]
        q = (q + rand_seq(rng, m))[:m]
        idx = {65: 0, 67: 1, 71: 2, 84: 3}
        p1 = np.zeros((6, m), dtype=np.float32)
        for jx, ch in enumerate(q):
            col = rng.random(4).astype(np.float32) * np.float32(0.2)
            col[idx.get(ch, 0)] += np.float32(1.0)
            p1[:4, jx] = col / col.sum()
        p2 = orc.create_profile_str(ref)
        for rc in (False, True):
            oriented = orc.revcomp_profile(p2) if rc else p2
            want = orc.gotoh_prof(p1, oriented, 1, 0, SC)
            got = emu.run(p1, ref, SC, 1, 0, emu.MODE_QP, K, revcomp=rc, band=W)
            assert got[2] == 0
            # the bound's top
            x = (SC[0] - SC[1]) * p1[:4].max(axis=0).astype(np.float64) + SC[1] * p1[:5].sum(axis=0).astype(np.float64)
            top = float(np.maximum(np.floor(x + 1e-4), 0).sum())  # (never below the kernel's truncated float chain)
            if got[0] > top - (-SC[3]) * (W + 1):
                certified += 1
                assert (got[0], got[1]) == want, (m, n, W, rc)
            else:
                assert got[0] <= want[0], (m, n, W, rc)
    assert certified >= 6
