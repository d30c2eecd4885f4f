"""GPU parity: the HIP kernels, called through the C ABI, against the oracle (bit exact)."""
import os

import numpy as np
import pytest

import pyoracle as orc

pytestmark = pytest.mark.gpu

SC = (3, -5, -10, -4)
CONFIGS = [(0, 0), (1, 0), (0, 1), (1, 1)]


@pytest.fixture(scope="module")
def ctx():
    import tracy_amd
    c = tracy_amd.Context(0)
    yield c
    c.close()


def rand_seq(rng, n, alpha=b"ACGT"):
    return bytes(rng.choice(list(alpha), size=n).tolist())


def rand_profile(rng, n, sharp=True):
    p = np.zeros((6, n), dtype=np.float32)
    x = rng.random((4, n)).astype(np.float32)
    if sharp:
        x = x ** 6
    p[:4] = x / x.sum(axis=0, keepdims=True)
    return p


def noisy_copy(rng, s, rate=0.05):
    out = bytearray()
    for ch in s:
        u = rng.random()
        if u < rate / 3:
            continue
        if u < 2 * rate / 3:
            out.append(int(rng.choice(list(b"ACGT"))))
        out.append(int(rng.choice(list(b"ACGT"))) if u > 1 - rate / 3 else ch)
    return bytes(out)


@pytest.mark.parametrize("cfg", CONFIGS)
def test_gotoh_char_ragged_batch(ctx, cfg):
    rng = np.random.default_rng(1 + cfg[0] * 2 + cfg[1])
    sizes = [(0, 0), (0, 7), (9, 0), (1, 1), (3, 200), (200, 3), (64, 64), (65, 130), (255, 257), (256, 300),
             (257, 100), (511, 513), (700, 90), (1024, 300), (1025, 64), (1500, 200)]
    a1, a2 = [], []
    for (m, n) in sizes:
        ref = rand_seq(rng, n, b"AC" if (m + n) % 2 else b"ACGT")
        q = noisy_copy(rng, (ref * (m // max(n, 1) + 1))[:m]) if n else rand_seq(rng, m)
        q = (q + rand_seq(rng, m))[:m]
        a1.append(q)
        a2.append(ref)
    params = SC + cfg
    scores, btr, rows = ctx.align(a1, a2, params, rows=True)
    sc_only = ctx.score(a1, a2, params)
    for i, (q, r) in enumerate(zip(a1, a2)):
        want = orc.gotoh_str(q, r, cfg[0], cfg[1], SC)
        assert (int(scores[i]), btr[i]) == want, (sizes[i], cfg)
        assert int(sc_only[i]) == want[0]
        assert rows[i] == orc.create_alignment_str(want[1], q, r)


def test_gotoh_profile_vs_string_reference(ctx):
    """gotoh(trace profile, _createProfile(reference)) -- the `tracy align` call shape (sage.h:239-258)"""
    rng = np.random.default_rng(5)
    profs, refs = [], []
    for (m, n) in [(1, 30), (40, 400), (300, 1200), (900, 2000), (1000, 1100), (1100, 700)]:
        profs.append(rand_profile(rng, m))
        refs.append(rand_seq(rng, n, b"ACGTACGTACGTACGTNn-x"))
    for cfg in [(1, 0), (1, 1), (0, 0)]:
        scores, btr, rows = ctx.align(profs, refs, SC + cfg, rows=True)
        sc_only = ctx.score(profs, refs, SC + cfg)
        for i in range(len(profs)):
            p2 = orc.create_profile_str(refs[i])
            want = orc.gotoh_prof(profs[i], p2, cfg[0], cfg[1], SC)
            assert (int(scores[i]), btr[i]) == want
            assert int(sc_only[i]) == want[0]
            assert rows[i] == orc.create_alignment_prof(want[1], profs[i], p2)


def test_gotoh_profile_profile(ctx):
    rng = np.random.default_rng(6)
    p1 = [rand_profile(rng, m, sharp=False) for m in (5, 100, 513, 600)]
    p2 = [rand_profile(rng, n) for n in (9, 333, 200, 700)]
    # rows 4 ('N') / 5 ('-') carry weight in alignment profiles: pairs 1 and 2 take the 25-term path, 0 and 3 the
    # 16-term path (row 4 zero in both profiles)
    p1[1][4, 50] = np.float32(0.2)
    p1[1][5, 51] = np.float32(0.4)
    p2[2][4, 199] = np.float32(1.0)
    for cfg in [(1, 0), (1, 1)]:
        scores, btr, rows = ctx.align(p1, p2, SC + cfg, rows=True)
        sc_only = ctx.score(p1, p2, SC + cfg)
        for i in range(len(p1)):
            want = orc.gotoh_prof(p1[i], p2[i], cfg[0], cfg[1], SC)
            assert (int(scores[i]), btr[i]) == want
            assert int(sc_only[i]) == want[0]
            assert rows[i] == orc.create_alignment_prof(want[1], p1[i], p2[i])


def special_profile(rng, n, kinds, nt=4):
    """profile whose columns mix the classes the screened score treats differently (test_emu_wave.screen_columns)"""
    from test_emu_wave import screen_columns
    cols = np.concatenate([screen_columns(rng, n, kd, nt) for kd in kinds])
    cols = cols[rng.permutation(len(cols))[:n]]
    return np.ascontiguousarray(np.vstack([cols.T, np.zeros((1, n), np.float32)]))


def test_profile_profile_screened_score(ctx, monkeypatch):
    """profile x profile substitution scores come from the screened short form where it is proven, from per-row tables
    against one-hot / uniform columns, and from the float chain otherwise (dp_kernels.h SubProf::screen): the oracle's
    result in every mix, and the same as with screening switched off"""
    import tracy_amd
    rng = np.random.default_rng(77)
    p1 = [special_profile(rng, 300, ["trace"]), special_profile(rng, 700, ["trace", "onehot", "uniform"]),
          special_profile(rng, 120, ["onehot"]), special_profile(rng, 256, ["dyadic", "trace"]),
          special_profile(rng, 90, ["consensus", "onehot"], 5), special_profile(rng, 1, ["trace"]),
          special_profile(rng, 400, ["heavy", "trace"])]
    p2 = [special_profile(rng, 420, ["trace"]), special_profile(rng, 500, ["trace", "onehot", "uniform", "dyadic"]),
          special_profile(rng, 333, ["onehot", "uniform"]), special_profile(rng, 64, ["dyadic"]),
          special_profile(rng, 210, ["consensus", "trace", "onehot"], 5), special_profile(rng, 70, ["onehot", "trace"]),
          special_profile(rng, 300, ["heavy", "onehot"])]
    ctx.set_option("no_screen", 1)
    plain = tracy_amd.Context(0)
    ctx.set_option("no_screen", 0)
    try:
        for sc in (SC, (5, -4, -6, -1)):
            for cfg in [(1, 0), (1, 1)]:
                scores, btr = ctx.align(p1, p2, sc + cfg)
                sc_only = ctx.score(p1, p2, sc + cfg)
                ref_scores, ref_btr = plain.align(p1, p2, sc + cfg)
                assert np.array_equal(scores, ref_scores) and btr == ref_btr
                assert np.array_equal(sc_only, plain.score(p1, p2, sc + cfg))
                for i in range(len(p1)):
                    want = orc.gotoh_prof(p1[i], p2[i], cfg[0], cfg[1], sc)
                    assert (int(scores[i]), btr[i]) == want, (i, sc, cfg)
                    assert int(sc_only[i]) == want[0]
    finally:
        plain.close()


def test_needle(ctx):
    sc = (5, -4, -10, -1)
    rng = np.random.default_rng(8)
    a1 = [rand_seq(rng, m, b"AC") for m in (0, 4, 90, 300, 1100)]
    a2 = [rand_seq(rng, n, b"AC") for n in (5, 0, 77, 310, 400)]
    for cfg in CONFIGS:
        scores, btr = ctx.align(a1, a2, sc + cfg, needle=True)
        sc_only = ctx.score(a1, a2, sc + cfg, needle=True)
        for i in range(len(a1)):
            want = orc.needle_str(a1[i], a2[i], cfg[0], cfg[1], sc)
            assert (int(scores[i]), btr[i]) == want
            assert int(sc_only[i]) == want[0]
    p1, p2 = [rand_profile(rng, 120, sharp=False)], [rand_profile(rng, 333, sharp=False)]
    scores, btr = ctx.align(p1, p2, sc + (1, 1), needle=True)
    assert (int(scores[0]), btr[0]) == orc.needle_prof(p1[0], p2[0], 1, 1, sc)


def test_all_pairs_indexing_and_chunking(ctx):
    """shared sequences through index arrays; a tiny workspace limit forces several traceback chunks"""
    rng = np.random.default_rng(9)
    seqs = [rand_seq(rng, int(rng.integers(50, 400))) for _ in range(12)]
    idx1 = [i for i in range(12) for j in range(12) if i < j]
    idx2 = [j for i in range(12) for j in range(12) if i < j]
    ctx.set_workspace_limit(2 << 20)
    try:
        scores, btr = ctx.align(seqs, seqs, SC + (1, 1), idx1=idx1, idx2=idx2)
    finally:
        ctx.set_workspace_limit(0)
    for k, (i, j) in enumerate(zip(idx1, idx2)):
        assert (int(scores[k]), btr[k]) == orc.gotoh_str(seqs[i], seqs[j], 1, 1, SC)


def test_traceback_walked_by_the_sweeping_workgroup(ctx, monkeypatch):
    """the fused walk (default) and the walk launch of its own give the same ops: ragged strings and profiles, single- and
    multi-pass strips, chunked workspaces, all four AlignConfigs"""
    rng = np.random.default_rng(31)
    sizes = [(0, 5), (5, 0), (1, 1), (63, 64), (64, 63), (300, 1200), (1025, 64), (1500, 300), (900, 1000), (17, 2000)]
    a1 = [rand_seq(rng, m) for m, _ in sizes]
    a2 = [noisy_copy(rng, (q * (n // max(len(q), 1) + 1))[:n]) if q else rand_seq(rng, n) for q, (_, n) in zip(a1, sizes)]
    profs = [rand_profile(rng, m) for m in (40, 333, 1100)]
    refs = [rand_seq(rng, n, b"ACGTN") for n in (700, 50, 1300)]
    for cfg in CONFIGS:
        ctx.set_workspace_limit(8 << 20 if cfg == CONFIGS[0] else 0)
        try:
            fused = ctx.align(a1, a2, SC + cfg, rows=True)
            fused_p = ctx.align(profs, refs, SC + cfg)
            ctx.set_option("no_fused_walk", 1)
            apart = ctx.align(a1, a2, SC + cfg, rows=True)
            apart_p = ctx.align(profs, refs, SC + cfg)
            ctx.set_option("no_fused_walk", 0)
        finally:
            ctx.set_workspace_limit(0)
        assert [int(x) for x in fused[0]] == [int(x) for x in apart[0]] and fused[1] == apart[1] and fused[2] == apart[2]
        assert [int(x) for x in fused_p[0]] == [int(x) for x in apart_p[0]] and fused_p[1] == apart_p[1]
        for i, (q, r) in enumerate(zip(a1, a2)):
            assert (int(fused[0][i]), fused[1][i]) == orc.gotoh_str(q, r, cfg[0], cfg[1], SC)


def test_long_pair_list(ctx):
    """pair lists of 2^16 pairs and more are filled and laid out by several host threads (descriptors, boundary-row scratch of
    multi-pass pairs as a scan over the threads' totals): the same scores as the same list handed over in two shorter calls,
    and the oracle's on a sample"""
    rng = np.random.default_rng(4711)
    seqs = [rand_seq(rng, int(rng.integers(1, 40))) for _ in range(372)]
    seqs[5] = rand_seq(rng, 1100)  # more rows than one pass of the tallest strip holds: its pairs use the scratch rows
    seqs[200] = rand_seq(rng, 1300)
    seqs[371] = b""
    idx1 = np.array([i for i in range(372) for j in range(372) if i < j], dtype=np.uint32)
    idx2 = np.array([j for i in range(372) for j in range(372) if i < j], dtype=np.uint32)
    assert len(idx1) >= (1 << 16)
    for params in (SC + (1, 1), SC + (1, 0)):
        whole = ctx.score(seqs, seqs, params, idx1=idx1, idx2=idx2)
        h = len(idx1) // 2
        parts = np.concatenate([ctx.score(seqs, seqs, params, idx1=idx1[:h], idx2=idx2[:h]), ctx.score(seqs, seqs, params, idx1=idx1[h:], idx2=idx2[h:])])
        assert np.array_equal(whole, parts)
        sample = list(rng.integers(0, len(idx1), 300)) + [k for k in range(len(idx1)) if idx1[k] in (5, 200) or idx2[k] in (5, 200, 371)][:400]
        for k in sample:
            assert int(whole[k]) == orc.gotoh_score_str(seqs[idx1[k]], seqs[idx2[k]], params[4], params[5], SC), k


def test_error_reporting(ctx):
    import tracy_amd
    with pytest.raises(tracy_amd.TracyHipError) as e:
        ctx.score([b"ACGT"], [b"ACGT"], (50000, -5, -10, -4, 1, 0))
    assert e.value.code == tracy_amd.capi.ERR_RANGE
    rng = np.random.default_rng(3)
    big = rand_profile(rng, 20) * np.float32(1e6)
    with pytest.raises(tracy_amd.TracyHipError) as e:
        ctx.score([big], [b"ACGTACGT"], SC + (1, 0))
    assert e.value.code == tracy_amd.capi.ERR_RANGE


def synth_trace_profile(rng, ref, mf, reverse, noise=0.02):
    """profile of a trace copied from `ref` (or its reverse complement) with substitutions/indels"""
    from sage_oracle import revcomp
    src = revcomp(ref) if reverse else ref
    start = int(rng.integers(0, max(1, len(src) - mf)))
    seq = noisy_copy(rng, src[start:start + mf + 20], noise)[:mf]
    seq = (seq + rand_seq(rng, mf))[:mf]
    p = np.zeros((6, mf), dtype=np.float32)
    idx = {65: 0, 67: 1, 71: 2, 84: 3}
    for j, ch in enumerate(seq):
        if ch not in idx:
            ch = int(rng.choice(list(b"ACGT")))
        main = np.float32(rng.uniform(0.7, 1.0))
        rest = rng.random(3).astype(np.float32)
        rest = rest / rest.sum() * (np.float32(1) - main)
        col = np.zeros(4, dtype=np.float32)
        col[idx[ch]] = main
        col[[k for k in range(4) if k != idx[ch]]] = rest
        p[:4, j] = col
    return p


def test_align_traces_pipeline(ctx):
    """tracyhip_align_traces == sage.h:191-311 composed from the oracle"""
    from sage_oracle import align_trace
    rng = np.random.default_rng(42)
    refs, profs = [], []
    for i, (mf, n) in enumerate([(300, 900), (420, 1500), (260, 700), (1000, 2500), (150, 400), (90, 300)]):
        ref = rand_seq(rng, n, b"ACGTACGTACGTACGTN")
        refs.append(ref)
        profs.append(synth_trace_profile(rng, ref, mf, reverse=bool(i % 2)))
    got = ctx.align_traces(profs, refs, SC, 50, 50)
    for i in range(len(refs)):
        want = align_trace(profs[i], refs[i], SC, 50, 50)
        for k in ("score_fwd", "score_rev", "forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
            assert int(got[k][i]) == int(want[k]), (i, k)
        assert got["btr"][i] == want["btr"], i
    assert sorted(got["forward"].tolist()) == [0, 0, 0, 1, 1, 1]


def test_align_traces_fallback_paths(ctx):
    """traces longer than one pass (mf > 1024 + trims) and scoring outside the band/16-bit domain (ge = 0)
    take the int32 + full-matrix traceback path of the same pipeline"""
    from sage_oracle import align_trace
    rng = np.random.default_rng(43)
    refs, profs = [], []
    for i, (mf, n) in enumerate([(1300, 2600), (1150, 1900), (400, 1200)]):
        ref = rand_seq(rng, n, b"ACGT")
        refs.append(ref)
        profs.append(synth_trace_profile(rng, ref, mf, reverse=bool(i % 2)))
    for sc in (SC, (3, -5, -10, 0), (2, -3, -2, -1)):
        got = ctx.align_traces(profs, refs, sc, 50, 50)
        for i in range(len(refs)):
            want = align_trace(profs[i], refs[i], sc, 50, 50)
            for k in ("score_fwd", "score_rev", "forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
                assert int(got[k][i]) == int(want[k]), (sc, i, k)
            assert got["btr"][i] == want["btr"], (sc, i)
    # mixed batch: short traces ride the band path, the batch as a whole falls back when one trace needs two passes
    got = ctx.align_traces(profs[2:], refs[2:], SC, 50, 50)
    want = align_trace(profs[2], refs[2], SC, 50, 50)
    assert got["btr"][0] == want["btr"] and int(got["score_final"][0]) == want["score_final"]


def test_cpp_host_mirror(tmp_path):
    """tracy_amd/host/tracy_amd.hpp keeps the reference's call shape: gotoh(a1, a2, align, ac, sc)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_mirror")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(root, "tests/cpp/test_mirror.cpp"),
                           "-L" + os.path.join(root, "tracy_amd/lib"), "-ltracy_hip", "-ltracy_host", "-L" + os.path.join(root, "oracle"),
                           "-ltracy_oracle", "-Wl,-rpath," + os.path.join(root, "tracy_amd/lib"),
                           "-Wl,-rpath," + os.path.join(root, "oracle"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_align_traces_against_the_largest_fasta_window(ctx):
    """MAX_SINGLE_FASTA_SIZE (fasta.h:10-12): 50 kbp references, forward and reverse traces, through the checkpointed
    score pass + band traceback"""
    import sage_oracle as so
    from tracy_amd import hostlib
    refs, profs, rev = hostlib.synth_align(31, 3, 50000, 800, 2)
    got = ctx.align_traces(list(profs), [r.tobytes() for r in refs], SC, 50, 50)
    for i in range(3):
        want = so.align_trace(profs[i], refs[i].tobytes(), SC, 50, 50)
        for k in ("score_fwd", "score_rev", "forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
            assert int(got[k][i]) == int(want[k]), (i, k)
        assert got["btr"][i] == want["btr"]
        assert int(got["forward"][i]) == 1 - int(rev[i])


def test_strand_by_certificate(ctx):
    """default mode of tracyhip_align_traces: the losing strand may be represented by a certified upper bound of its
    score (prefix pass + row maxima); strand, slice, preliminary and final alignment are those of the reference, the
    winner's score is exact and the loser's entry bounds its true score from above without reaching the winner's"""
    import sage_oracle as so
    from tracy_amd import hostlib
    refs, profs, rev = hostlib.synth_align(8100, 24, 4000, 700, 2)
    profs = list(profs)
    # two hard cases: a trace unrelated to its window (bounds close together) and a noisy trace
    rng = np.random.default_rng(3)
    profs[5] = rand_profile(rng, 700)
    noisy = profs[6].copy()
    noisy[:4] = 0.6 * noisy[:4] + 0.4 * rand_profile(rng, 700, sharp=False)[:4]
    profs[6] = np.ascontiguousarray(noisy / noisy[:4].sum(axis=0, keepdims=True))
    refl = [r.tobytes() for r in refs]
    exact = ctx.align_traces(profs, refl, SC, 50, 50, exact_scores=True)
    ctx.set_option("no_vote", 1)  # the two-stage form: prefix bounds of both strands, then the likely winner
    try:
        fast2 = ctx.align_traces(profs, refl, SC, 50, 50, exact_scores=False)
    finally:
        ctx.set_option("no_vote", 0)
    fast = ctx.align_traces(profs, refl, SC, 50, 50, exact_scores=False)  # k-mer vote + prefix bounds in the sweep launch
    for k in ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
        assert np.array_equal(np.asarray(fast2[k]), np.asarray(exact[k])), k
    assert fast2["btr"] == exact["btr"]
    for i in range(24):
        wn, ls = ("score_fwd", "score_rev") if int(exact["forward"][i]) else ("score_rev", "score_fwd")
        assert int(fast2[wn][i]) == int(exact[wn][i]) and int(fast2[ls][i]) >= int(exact[ls][i])
    nb = 0
    for i in range(24):
        want = so.align_trace(profs[i], refl[i], SC, 50, 50)
        for k in ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
            assert int(fast[k][i]) == int(want[k]) == int(exact[k][i]), (i, k)
        assert fast["btr"][i] == want["btr"] == exact["btr"][i]
        assert int(exact["score_fwd"][i]) == want["score_fwd"] and int(exact["score_rev"][i]) == want["score_rev"]
        w, l = ("score_fwd", "score_rev") if want["forward"] else ("score_rev", "score_fwd")
        assert int(fast[w][i]) == want[w]
        assert int(fast[l][i]) >= want[l]
        assert (int(fast[l][i]) < int(fast[w][i])) if want["forward"] else (int(fast[l][i]) <= int(fast[w][i]))
        nb += int(fast[l][i]) != want[l]
    assert nb >= 12  # most losers were decided by their bound


def test_align_traces_lanes(ctx):
    """tracyhip_set_lanes: the batch split over concurrent chunks gives the arrays of the single-lane call, in host and
    in shared-reference (ref_index) form, in both orientation modes"""
    import tracy_amd
    from tracy_amd import hostlib
    nt = 300
    refs, profs, rev = hostlib.synth_align(9100, nt, 1500, 420, 2)
    profs = list(profs)
    refl = [r.tobytes() for r in refs]
    c3 = tracy_amd.Context(0)
    c3.set_lanes(3)
    try:
        for exact in (True, False):
            one = ctx.align_traces(profs, refl, SC, 50, 50, exact_scores=exact)
            many = c3.align_traces(profs, refl, SC, 50, 50, exact_scores=exact)
            for k in ("score_fwd", "score_rev", "forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
                assert np.array_equal(np.asarray(one[k]), np.asarray(many[k])), (exact, k)
            assert one["btr"] == many["btr"]
        # one shared reference for every trace (the FASTA case of the CLI)
        ridx = np.zeros(nt, dtype=np.uint32)
        one = ctx.align_traces(profs, refl[:1], SC, 50, 50, ref_index=ridx)
        many = c3.align_traces(profs, refl[:1], SC, 50, 50, ref_index=ridx)
        for k in ("score_fwd", "score_rev", "forward", "score_final", "slice_begin", "slice_len"):
            assert np.array_equal(np.asarray(one[k]), np.asarray(many[k])), k
        assert one["btr"] == many["btr"]
        # an invalid reference is reported from a lane as it is from the single context
        bad = list(refl)
        bad[nt - 1] = bad[nt - 1][:-1] + b"x"
        with pytest.raises(RuntimeError, match="upper-case"):
            c3.align_traces(profs, bad, SC, 50, 50)
    finally:
        c3.set_lanes(1)
