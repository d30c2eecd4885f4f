"""GPU parity at the edges of the kernels' value ranges: the C ABI takes arbitrary float profiles and scorings, the
reference (align.h:103-118, gotoh.h:124-139) computes in int32 without limits.  Every call here must return either the
oracle's exact result or TRACYHIP_ERR_RANGE -- never a wrapped score.  The 16-bit score sweeps and the 14-bit origin
field rest on bounds that hold for NORMALISED profiles; kernels report larger query-profile entries / column masses and
the host re-checks the range with the real bound (capi.hip range_verdict), repeating the work on the int32 kernels."""
import numpy as np
import pytest

import pyoracle as orc

pytestmark = pytest.mark.gpu

SC = (3, -5, -10, -4)


@pytest.fixture(scope="module")
def ctx():
    import tracy_amd
    c = tracy_amd.Context(0)
    yield c
    c.close()


def rand_seq(rng, n, alpha=b"ACGT"):
    return bytes(rng.choice(list(alpha), size=n).tolist())


def profile_of(rng, seq, scale=1.0, noise=0.15):
    """a trace-like profile of `seq`: the called base dominates; column sums = scale"""
    n = len(seq)
    p = np.zeros((6, n), dtype=np.float32)
    x = (rng.random((4, n)) * noise).astype(np.float32)
    idx = np.array([b"ACGT".index(c) for c in seq])
    x[idx, np.arange(n)] += 1.0
    p[:4] = x / x.sum(axis=0, keepdims=True) * np.float32(scale)
    return p


def exact_or_range(call, want, must_be_exact):
    """run `call`; it has to produce `want` or raise ERR_RANGE (only allowed when must_be_exact is False)"""
    from tracy_amd.capi import TracyHipError, ERR_RANGE
    try:
        got = call()
    except TracyHipError as e:
        assert e.code == ERR_RANGE, e
        assert not must_be_exact, "ERR_RANGE where the int32 kernels hold the values: %s" % e
        return "range"
    assert got == want, (got, want)
    return "exact"


def noisy_window(rng, seq, n):
    """a reference window of n bases that contains seq (with a few edits) somewhere inside"""
    m = len(seq)
    s = bytearray(seq)
    for _ in range(max(1, m // 60)):
        k = int(rng.integers(0, m))
        s[k] = int(rng.choice(list(b"ACGT")))
    lead = int(rng.integers(0, n - m + 1)) if n > m else 0
    return (rand_seq(rng, lead) + bytes(s) + rand_seq(rng, n))[:n]


@pytest.mark.parametrize("scale", [2.0, 8.0, 12.0, 100.0, 1000.0, 20000.0])
def test_unnormalised_profile_vs_string(ctx, scale):
    """column sums 2 .. 20000: rows x |q| leaves the int16 range of the 16-bit sweep long before it leaves int32"""
    rng = np.random.default_rng(int(scale))
    profs, refs = [], []
    for (m, n) in [(900, 2400), (960, 1500), (300, 1200), (64, 90)]:
        seq = rand_seq(rng, m)
        profs.append(profile_of(rng, seq, scale))
        refs.append(noisy_window(rng, seq, n))
    outcomes = []
    for cfg in [(1, 0), (0, 0), (1, 1)]:
        want_sc, want_al = [], []
        for p, r in zip(profs, refs):
            sc, btr = orc.gotoh_prof(p, orc.create_profile_str(r), cfg[0], cfg[1], SC)
            want_sc.append(sc)
            want_al.append((sc, btr))
        # score-only: exact as long as the largest entry fits the int16 table (|q| <= 32767: scale * 5 here)
        outcomes.append(exact_or_range(lambda: [int(x) for x in ctx.score(profs, refs, SC + cfg)], want_sc, scale * 5 <= 30000))
        # traceback kernels keep scores x32 in the int16 table: |q| <= 1023
        def aligned():
            s, b = ctx.align(profs, refs, SC + cfg)
            return [(int(x), y) for x, y in zip(s, b)]
        outcomes.append(exact_or_range(aligned, want_al, scale * 5 <= 1000))
    if scale >= 20000.0:
        assert "range" in outcomes


@pytest.mark.parametrize("scale", [2.0, 12.0, 100.0])
def test_unnormalised_profiles_through_align_traces(ctx, scale):
    """the `tracy align` pipeline: the checkpointed 16-bit sweep must notice the profile and restart on the int32 kernels"""
    from sage_oracle import align_trace
    rng = np.random.default_rng(100 + int(scale))
    profs, refs = [], []
    for t in range(6):
        m = int(rng.integers(700, 960))
        seq = rand_seq(rng, m)
        ref = noisy_window(rng, seq, int(rng.integers(1500, 2600)))
        if t % 2:
            ref = ref[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
        profs.append(profile_of(rng, seq, scale))
        refs.append(ref)
    for exact in (True, False):
        got = ctx.align_traces(profs, refs, SC, 50, 50, exact_scores=exact)
        for i in range(len(profs)):
            want = align_trace(profs[i], refs[i], SC, 50, 50)
            for k in ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
                assert int(got[k][i]) == int(want[k]), (scale, exact, i, k)
            assert got["btr"][i] == want["btr"], (scale, exact, i)
            if exact:
                assert (int(got["score_fwd"][i]), int(got["score_rev"][i])) == (int(want["score_fwd"]), int(want["score_rev"]))


def test_negative_profile_entries(ctx):
    """createProfile can emit entries outside [0, 1] when ABIF samples are negative (normfac outside [0, 1], profile.h:44-50)"""
    rng = np.random.default_rng(7)
    m, n = 940, 2000
    seq = rand_seq(rng, m)
    p = profile_of(rng, seq, 1.0)
    shift = (rng.random(m) * 3.0).astype(np.float32)      # p_called + shift, the others - shift / 3: sums stay 1
    idx = np.array([b"ACGT".index(c) for c in seq])
    for k in range(4):
        p[k] += np.where(idx == k, shift, -shift / 3.0).astype(np.float32)
    ref = noisy_window(rng, seq, n)
    for cfg in [(1, 0), (0, 0)]:
        want = orc.gotoh_prof(p, orc.create_profile_str(ref), cfg[0], cfg[1], SC)
        assert int(ctx.score([p], [ref], SC + cfg)[0]) == want[0]
        s, b = ctx.align([p], [ref], SC + cfg)
        assert (int(s[0]), b[0]) == want


@pytest.mark.parametrize("match", [29, 30, 31, 32, 33, 40])
def test_scoring_around_the_int16_limit(ctx, match):
    """rows x match around 30 000: K = 16 strips hold 1024 rows; a perfect 1024-base hit scores 1024 x match (29 696 .. 40 960).
    Past the limit the 16-bit kernel is not eligible (narrow_ok) and the int32 kernel must deliver the same number."""
    rng = np.random.default_rng(match)
    sc = (match, -5, -10, -4)
    seqs = [rand_seq(rng, 1024), rand_seq(rng, 960), rand_seq(rng, 1000)]
    refs = [rand_seq(rng, 300) + s + rand_seq(rng, 200) for s in seqs]
    onehot = [orc.create_profile_str(s) for s in seqs]
    for a1 in (seqs, onehot):
        got = ctx.score(a1, refs, sc + (1, 0))
        for i in range(len(seqs)):
            if a1 is seqs:
                want = orc.gotoh_score_str(seqs[i], refs[i], 1, 0, sc)
            else:
                want = orc.gotoh_score_prof(onehot[i], orc.create_profile_str(refs[i]), 1, 0, sc)
            assert want == len(seqs[i]) * match
            assert int(got[i]) == want, (match, i)
    # and through the pipeline (checkpointed sweep, band traceback)
    from sage_oracle import align_trace
    profs = [orc.create_profile_str(s) for s in seqs]
    got = ctx.align_traces(profs, refs, sc, 10, 10, exact_scores=True)
    for i in range(len(seqs)):
        want = align_trace(profs[i], refs[i], sc, 10, 10)
        for k in ("score_fwd", "score_rev", "forward", "score_prelim", "slice_begin", "slice_len", "score_final"):
            assert int(got[k][i]) == int(want[k]), (match, i, k)
        assert got["btr"][i] == want["btr"]


def test_huge_gap_open_small_extension(ctx):
    """ge = -1 with go = -19 000: the sentinel arithmetic of the 16-bit kernel has no room, the int32 kernels do"""
    rng = np.random.default_rng(11)
    sc = (3, -5, -19000, -1)
    seq = rand_seq(rng, 900)
    ref = noisy_window(rng, seq, 1800)
    p = profile_of(rng, seq)
    for cfg in [(1, 0), (0, 0)]:
        want = orc.gotoh_prof(p, orc.create_profile_str(ref), cfg[0], cfg[1], sc)
        assert int(ctx.score([p], [ref], sc + cfg)[0]) == want[0]
        s, b = ctx.align([p], [ref], sc + cfg)
        assert (int(s[0]), b[0]) == want
        want = orc.gotoh_str(seq, ref, cfg[0], cfg[1], sc)
        s, b = ctx.align([seq], [ref], sc + cfg)
        assert (int(s[0]), b[0]) == want
    # a reference long enough to leave the x32 range of the traceback kernels: ERR_RANGE, not a wrong alignment
    long_ref = noisy_window(rng, seq, 12000)
    want = orc.gotoh_str(seq, long_ref, 1, 0, sc)
    def aligned():
        s, b = ctx.align([seq], [long_ref], sc + (1, 0))
        return (int(s[0]), b[0])
    exact_or_range(aligned, want, False)


@pytest.mark.parametrize("scale", [3.0, 40.0, 1.0e4])
def test_unnormalised_profile_x_profile(ctx, scale):
    """profile x profile (`assemble`): |score| <= mass(a) mass(b) max(|match|, |mismatch|)"""
    rng = np.random.default_rng(int(scale) + 3)
    sa, sb = rand_seq(rng, 400), rand_seq(rng, 520)
    p1 = [profile_of(rng, sa, scale, noise=0.6)]
    p2 = [profile_of(rng, sb[:100] + sa[50:350] + sb[100:220], scale, noise=0.6)]
    for cfg in [(1, 1), (1, 0)]:
        want = orc.gotoh_prof(p1[0], p2[0], cfg[0], cfg[1], SC)
        exact_or_range(lambda: int(ctx.score(p1, p2, SC + cfg)[0]), want[0], scale <= 40.0)
        def aligned():
            s, b = ctx.align(p1, p2, SC + cfg)
            return (int(s[0]), b[0])
        exact_or_range(aligned, want, scale <= 40.0)


def test_nan_profile_is_a_range_error(ctx):
    from tracy_amd.capi import TracyHipError, ERR_RANGE
    rng = np.random.default_rng(5)
    p1 = [profile_of(rng, rand_seq(rng, 200))]
    p2 = [profile_of(rng, rand_seq(rng, 260))]
    p2[0][2, 17] = np.float32("nan")
    with pytest.raises(TracyHipError) as ei:
        ctx.score(p1, p2, SC + (1, 1))
    assert ei.value.code == ERR_RANGE


@pytest.mark.parametrize("score", [(2000, -3000, -5000, -2500), (1500, -1001, -10, -4), (7, -9, -12000, -1)])
def test_scorings_beyond_a_thousand(ctx, score):
    """|match|, |mismatch| > 1000: x 32 they leave the int16 tables of the tagged tracebacks, the band kernels and the 16-bit sweeps;
    strings take the byte-compare kernels, profile rows the tracebacks with an unshifted table, the pipelines their whole-matrix
    forms -- the oracle's exact result, as the reference's int arithmetic gives it for any scoring (align.h:11-32)"""
    from tracy_amd import hostlib
    from sage_oracle import align_trace
    rng = np.random.default_rng(abs(score[0]) + 1)
    # generic entry points
    a = [rand_seq(rng, int(rng.integers(1, 400))) for _ in range(24)]
    b = [noisy_window(rng, x, len(x) + int(rng.integers(0, 300))) for x in a]
    for hfree in (1, 0):
        sc, btr = ctx.align(a, b, score + (hfree, 0))
        for i in range(len(a)):
            assert (int(sc[i]), btr[i]) == orc.gotoh_str(a[i], b[i], hfree, 0, score), (i, hfree)
    profs = [profile_of(rng, x) for x in a]
    sc, btr = ctx.align(profs, b, score + (1, 0))
    s2 = ctx.score(profs, b, score + (1, 0))
    for i in range(len(a)):
        want = orc.gotoh_prof(profs[i], orc.create_profile_str(b[i]), 1, 0, score)
        assert (int(sc[i]), btr[i]) == want and int(s2[i]) == want[0], i
    # the pipelines
    refs, tprofs, rev = hostlib.synth_align(55, 6, 1500, 600, 0)
    refl = [r.tobytes() for r in refs]
    got = ctx.align_traces(list(tprofs), refl, score, 50, 50)
    for i in range(6):
        want = align_trace(tprofs[i], refl[i], score, 50, 50)
        for k in ("score_fwd", "score_rev", "forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
            assert int(got[k][i]) == int(want[k]), (i, k)
        assert got["btr"][i] == want["btr"], i


def test_decompose_with_a_wide_scoring(ctx):
    from indigo_oracle import decompose_trace
    from tracy_amd import capi, hostlib
    score = (1200, -2000, -4000, -1600)
    nt = 6
    d = hostlib.synth_decompose_batch(777, nt, 1500, 500, 0, mix=1)
    sig = [d["signal"][i] for i in range(nt)]
    pos = [d["bcpos"][i] for i in range(nt)]
    pri = [d["primary"][i].tobytes() for i in range(nt)]
    sec = [d["secondary"][i].tobytes() for i in range(nt)]
    refs = [d["refs"][i].tobytes() for i in range(nt)]
    hbc = capi.HostBaseCalls(sig, pos, pri, sec)
    got = ctx.decompose_traces([d["profiles"][i] for i in range(nt)], hbc, refs, score)
    for i in range(nt):
        w = decompose_trace(sig[i], pos[i], pri[i], sec[i], refs[i], score)
        assert int(got["status"][i]) == w["status"], i
        if w["status"] != 0:
            continue
        assert got["primary"][i] == w["primary"] and got["secdecomp_list"][i] == w["secdecomp"] and got["dcp"][i] == w["dcp"], i
        for k in range(3):
            assert int(got["score%d" % k][i]) == w["score%d" % k] and got["btr%d" % k][i] == w["btr%d" % k], (i, k)
