"""GPU parity at the shapes BASELINE.json quotes its metric on, through the batch entry points of the C ABI:
configs[1]  1 kb traces `align` vs 10 kb windows (sage.h:191-311), both strands, both orientation modes;
configs[2]  1 kb heterozygous traces `decompose` vs 3 kb windows (indigo.h:190-388): het insertion / het deletion /
            homozygous indel / no variant, both strands;
configs[4]  900 x 900 profile x profile `gotohScore<true,true>` (msa.h:33-42) and its traceback.
Bit-exact against the oracle; sizes the oracle finishes in seconds per case."""
from concurrent.futures import ThreadPoolExecutor

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


@pytest.mark.parametrize("lanes", [1, 2])
def test_align_1kb_vs_10kb(lanes):
    import tracy_amd
    from tracy_amd import hostlib
    from sage_oracle import align_trace
    nt = 16 if lanes == 1 else 128  # lanes split batches of >= 64 traces per lane
    refs, profs, rev = hostlib.synth_align(4242, nt, 10000, 1000, 0)
    assert 0 < int(rev.sum()) < nt  # both strands present
    c = tracy_amd.Context(0)
    c.set_lanes(lanes)
    refl = [r.tobytes() for r in refs]
    exact = c.align_traces(list(profs), refl, SC, 50, 50, exact_scores=True)
    fast = c.align_traces(list(profs), refl, SC, 50, 50, exact_scores=False)
    c.close()
    check = range(nt) if lanes == 1 else list(range(0, nt, 9)) + [63, 64, 127]
    with ThreadPoolExecutor(max_workers=16) as pool:
        wants = list(pool.map(lambda i: align_trace(profs[i], refl[i], SC, 50, 50), check))
    for i, want in zip(check, wants):
        assert int(want["forward"]) == 1 - int(rev[i])
        for k in ("score_fwd", "score_rev", "forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
            assert int(exact[k][i]) == int(want[k]), (i, k)
        assert exact["btr"][i] == want["btr"], i
        for k in ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
            assert int(fast[k][i]) == int(want[k]), (i, k, "certificate")
        assert fast["btr"][i] == want["btr"], i
        win, lose = ("score_fwd", "score_rev") if want["forward"] else ("score_rev", "score_fwd")
        assert int(fast[win][i]) == int(want[win]) and int(fast[lose][i]) >= int(want[lose])


@pytest.mark.parametrize("with_short", [False, True])
def test_align_degenerate_traces_in_a_normal_batch(with_short):
    """traces whose optimal preliminary alignment is the all-gap path (row m never leaves its trailing run: c_e = 0) -- a poly-A
    one-hot trace against a 10 kb window over {C,G}, an all-N trace -- and a trace of one row, mixed into a normal batch:
    every trace equals the oracle (the reference aligns junk like anything else, gotoh.h:143-167, fmindex.h:429-463)"""
    import tracy_amd
    from tracy_amd import hostlib
    from sage_oracle import align_trace
    nt = 12
    refs, profs, rev = hostlib.synth_align(31337, nt, 10000, 1000, 0)
    refs = refs.copy()
    profl = [np.ascontiguousarray(p) for p in profs]
    polya = np.zeros((6, 1000), np.float32); polya[0] = 1.0
    alln = np.zeros((6, 1000), np.float32); alln[4] = 1.0
    rng = np.random.default_rng(5)
    refs[3] = np.frombuffer(b"CG", np.uint8)[rng.integers(0, 2, 10000)]  # no A on either strand
    profl[3] = polya
    profl[7] = alln
    profl[10] = polya.copy()   # poly-A against a normal window (it finds its A's: not degenerate)
    if with_short:             # ragged strip heights: the batch leaves the voted single-K path
        profl[5] = np.ascontiguousarray(profl[5][:, :1])
        profl[8] = np.ascontiguousarray(profl[8][:, :130])
    refl = [r.tobytes() for r in refs]
    c = tracy_amd.Context(0)
    try:
        for exact in (True, False):
            got = c.align_traces(profl, refl, SC, 50, 50, exact_scores=exact)
            with ThreadPoolExecutor(max_workers=16) as pool:
                wants = list(pool.map(lambda i: align_trace(profl[i], refl[i], SC, 50, 50), range(nt)))
            for i, want in enumerate(wants):
                for k in ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final") + (("score_fwd", "score_rev") if exact else ()):
                    assert int(got[k][i]) == int(want[k]), (i, k, exact)
                assert got["btr"][i] == want["btr"], (i, exact)
            for i in (3, 7):  # the all-gap alignment: n 'h' then m 'v', trimReferenceSlice keeps substr(0, trimRight)
                assert int(wants[i]["slice_begin"]) == 0 and int(wants[i]["slice_len"]) == 50, i
    finally:
        c.close()


def test_align_references_with_n_columns(monkeypatch):
    """the 16-bit sweeps and prefix bounds exist with a four-code table (references of A C G T) and a six-code one; the
    encoders' block map sends each pair to one of them.  A batch that mixes plain windows with windows holding N (at block
    borders, in the aligned region, as a run): the oracle's result, and the same with the compact forms switched off"""
    import tracy_amd
    from tracy_amd import hostlib
    from sage_oracle import align_trace
    nt = 24
    refs, profs, rev = hostlib.synth_align(777, nt, 4000, 1000, 0)
    refs = refs.copy()
    refs[1, 255] = ord("N"); refs[2, 256] = ord("N"); refs[3, 3999] = ord("N"); refs[4, 0] = ord("N")
    refs[5, 1500:1540] = ord("N"); refs[6, ::97] = ord("N"); refs[7, 2000] = ord("N")
    refl = [r.tobytes() for r in refs]
    c = tracy_amd.Context(0)
    monkeypatch.setenv("TRACYHIP_NO_COMPACT", "1")
    plain = tracy_amd.Context(0)
    monkeypatch.delenv("TRACYHIP_NO_COMPACT")
    try:
        for exact in (True, False):
            got = c.align_traces(list(profs), refl, SC, 50, 50, exact_scores=exact)
            ref = plain.align_traces(list(profs), refl, SC, 50, 50, exact_scores=exact)
            for k in ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final") + (("score_fwd", "score_rev") if exact else ()):
                assert np.array_equal(got[k], ref[k]), (k, exact)
            assert got["btr"] == ref["btr"]
            with ThreadPoolExecutor(max_workers=16) as pool:
                wants = list(pool.map(lambda i: align_trace(profs[i], refl[i], SC, 50, 50), range(10)))
            for i, want in enumerate(wants):
                for k in ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
                    assert int(got[k][i]) == int(want[k]), (i, k, exact)
                assert got["btr"][i] == want["btr"], (i, exact)
                if exact:
                    assert (int(got["score_fwd"][i]), int(got["score_rev"][i])) == (int(want["score_fwd"]), int(want["score_rev"]))
    finally:
        c.close()
        plain.close()


def test_decompose_1kb_vs_3kb(ctx):
    from tracy_amd import capi, hostlib
    import indigo_oracle as io
    nt = 20  # mix 1: traces 8, 18 homozygous indel, 9, 19 no variant, the rest het indels; odd traces on the reverse strand
    d = hostlib.synth_decompose_batch(90210, nt, 3000, 1000, 0, mix=1)
    sig = [d["signal"][i] for i in range(nt)]
    pos = [d["bcpos"][i] for i in range(nt)]
    pri = [d["primary"][i].tobytes() for i in range(nt)]
    sec = [d["secondary"][i].tobytes() for i in range(nt)]
    refs = [d["refs"][i].tobytes() for i in range(nt)]
    with ThreadPoolExecutor(max_workers=16) as pool:
        wants = list(pool.map(lambda i: io.decompose_trace(sig[i], pos[i], pri[i], sec[i], refs[i], SC), range(nt)))
    kinds = set()
    for exact in (True, False):
        hbc = capi.HostBaseCalls(sig, pos, pri, sec)
        got = ctx.decompose_traces([d["profiles"][i] for i in range(nt)], hbc, refs, SC, exact_scores=exact)
        fr = np.asarray(got["fractions"]).reshape(-1, 2)
        for i, w in enumerate(wants):
            assert int(got["forward"][i]) == w["forward"] == 1 - (i & 1), i
            assert int(got["status"][i]) == w["status"], i
            if w["status"] != 0:  # later outputs of failed traces are unspecified (include/tracy_hip.h)
                continue
            b = got["bp"][i]
            assert (b.indelshift, b.traceleft, b.breakpoint) == (w["bp"].indelshift, w["bp"].traceleft, w["bp"].breakpoint), i
            assert int(got["score_trim"][i]) == w["score_trim"]
            assert got["primary"][i] == w["primary"] and got["secondary"][i] == w["secondary"], i
            assert got["secdecomp_list"][i] == w["secdecomp"], i
            assert got["dcp"][i] == w["dcp"], i
            st = got["dstatus"][i]
            assert (st.kind, st.best_ins, st.best_del, st.best_fr) == tuple(w["dstatus"]), i
            assert (float(fr[i, 0]), float(fr[i, 1])) == w["af"], i
            for k in range(3):
                assert int(got["score%d" % k][i]) == w["score%d" % k], (i, k)
                assert got["btr%d" % k][i] == w["btr%d" % k], (i, k)
            for k in range(2):
                for nm in ("slice_begin", "slice_len", "ref_pos"):
                    assert int(got["%s%d" % (nm, k)][i]) == w["%s%d" % (nm, k)], (i, k, nm)
            if exact:
                assert (int(got["score_fwd"][i]), int(got["score_rev"][i])) == (w["score_fwd"], w["score_rev"])
            kinds.add((i % 10 == 8, i % 10 == 9, i & 1))
    assert len(kinds) >= 4


def test_profile_x_profile_900(ctx):
    """all-pairs shape of `tracy assemble`: trace profiles (row 4 zero: 16-term body) and alignment-like profiles with
    N / gap weight (25-term body), AlignConfig<true,true>"""
    from tracy_amd import hostlib
    refs, profs, rev = hostlib.synth_align(777, 8, 2000, 900, 0)
    p1 = [np.ascontiguousarray(profs[i]) for i in range(8)]
    p2 = [np.ascontiguousarray(profs[(i + 1) % 8]) for i in range(8)]
    # overlapping traces: the second half of one is the first half of the next (as tiled traces in `assemble`)
    for i in range(0, 8, 2):
        p2[i] = np.ascontiguousarray(np.concatenate([p1[i][:, 450:], p2[i][:, :450]], axis=1))
    for i in (1, 5):  # alignment-like columns
        p1[i] = p1[i].copy()
        p1[i][4, ::17] = np.float32(0.25)
        p1[i][5, ::29] = np.float32(0.125)
    sc_only = ctx.score(p1, p2, SC + (1, 1))
    scores, btr = ctx.align(p1, p2, SC + (1, 1))
    with ThreadPoolExecutor(max_workers=8) as pool:
        wants = list(pool.map(lambda i: orc.gotoh_prof(p1[i], p2[i], 1, 1, SC), range(8)))
    for i, w in enumerate(wants):
        assert int(sc_only[i]) == w[0], i
        assert (int(scores[i]), btr[i]) == w, i


def test_align_preliminary_alignment_by_its_two_ends(monkeypatch):
    """`tracy align` reads nothing of the preliminary alignment but trimReferenceSlice's two ends: by default an origin-tracking
    sweep over the certified sub-window delivers them (no checkpoints, no band traceback).  Same results as with the band
    traceback, in both orientation modes, also for traces that barely match their window (wide sub-windows) and for profiles
    whose entries exceed max(match, mismatch), for fuzzy profiles against windows with inserted segments; checked against the
    oracle's sage.h chain as well"""
    import tracy_amd
    from tracy_amd import hostlib
    nt = 40
    refs, profs, rev = hostlib.synth_align(2024, nt, 6000, 1000, 0)
    profs = profs.copy()
    rng = np.random.default_rng(5)
    profs[3] = rng.random(profs[3].shape).astype(np.float32)  # an unrelated trace: low score, wide sub-window
    profs[3][4:] = 0
    profs[3][:4] /= profs[3][:4].sum(axis=0, keepdims=True)
    profs[5][:4] *= np.float32(1.4)                           # column masses 1.4: scores above `match`
    # the sub-window is bounded by what the rows of the profile can score at best (not by match * rows): fuzzy columns make
    # that bound small, segments inserted into the window make the optimal path long -- the two have to stay consistent
    for t, (fuzz, ins) in {7: (0.6, 300), 8: (0.9, 700), 9: (0.3, 1500), 10: (0.97, 40), 11: (0.75, 2500)}.items():
        flat = np.zeros_like(profs[t])
        flat[:4] = 0.25
        profs[t] = ((1 - fuzz) * profs[t] + fuzz * flat).astype(np.float32)
        r = refs[t].copy()
        cut = 2500 + 60 * t
        seg = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), ins)
        refs[t] = np.concatenate([r[:cut], seg, r[cut:]])[:len(r)]
    refl = [r.tobytes() for r in refs]
    c = tracy_amd.Context(0)
    try:
        for exact in (True, False):
            c.set_option("no_prelim_origin", 0)
            got = c.align_traces(list(profs), refl, SC, 50, 50, exact_scores=exact)
            c.set_option("no_prelim_origin", 1)
            ref = c.align_traces(list(profs), refl, SC, 50, 50, exact_scores=exact)
            c.set_option("no_prelim_origin", 0)
            keys = ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final") + (("score_fwd", "score_rev") if exact else ())
            for k in keys:
                assert np.array_equal(got[k], ref[k]), (k, exact)
            assert got["btr"] == ref["btr"]
        import sage_oracle
        for t in (3, 7, 8, 9, 10, 11):
            want = sage_oracle.align_trace(profs[t], refl[t], SC, 50, 50)
            assert (int(got["slice_begin"][t]), int(got["slice_len"][t]), int(got["score_final"][t]), got["btr"][t]) == \
                   (want["slice_begin"], want["slice_len"], want["score_final"], want["btr"]), t
    finally:
        c.close()


def test_align_final_alignment_on_the_certified_band(monkeypatch):
    """The final alignments are a traceback DP on a diagonal band (TRACYHIP_BAND_W, default 48; 0 = whole matrices), certified per pair by score against the bound
    of the profile's row maxima, repeated on the whole matrix where the certificate fails.  Same results as the default path for
    a band that certifies (48), one that mostly does not (2: nearly every pair is repeated) and in between (12); against the
    oracle for the traces whose slices are longer / shorter than the trace or barely match"""
    import tracy_amd
    from tracy_amd import hostlib
    import sage_oracle
    nt = 24
    refs, profs, rev = hostlib.synth_align(4242, nt, 6000, 1000, 0)
    profs = profs.copy()
    rng = np.random.default_rng(8)
    profs[2] = rng.random(profs[2].shape).astype(np.float32)  # unrelated trace
    profs[2][4:] = 0
    profs[2][:4] /= profs[2][:4].sum(axis=0, keepdims=True)
    for t, (cut, ins) in {4: (2700, 60), 5: (3100, 200), 6: (2500, 9)}.items():  # segments inserted into / deleted from the window
        r = refs[t].copy()
        seg = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), ins)
        refs[t] = np.concatenate([r[:cut], seg, r[cut:]])[:len(r)]
    for t, (cut, dele) in {7: (2800, 40), 8: (3000, 150)}.items():
        r = refs[t].copy()
        refs[t] = np.concatenate([r[:cut], r[cut + dele:], r[:dele]])
    refl = [r.tobytes() for r in refs]
    keys = ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final", "score_fwd", "score_rev")
    c = tracy_amd.Context(0)
    try:
        c.set_option("band_w", 0)  # whole matrices
        ref = c.align_traces(list(profs), refl, SC, 50, 50, exact_scores=True)
        for wband in ("48", "12", "2", None):  # None: the default (and the stream-ordered pipeline)
            c.set_option("band_w", -1 if wband is None else wband)
            for lanes in (1, 2):
                c.set_lanes(lanes)
                got = c.align_traces(list(profs), refl, SC, 50, 50, exact_scores=True)
                for k in keys:
                    assert np.array_equal(got[k], ref[k]), (k, wband, lanes)
                assert got["btr"] == ref["btr"], (wband, lanes)
            c.set_lanes(1)
        c.set_option("band_w", -1)
        for t in (0, 2, 4, 5, 6, 7, 8):
            want = sage_oracle.align_trace(profs[t], refl[t], SC, 50, 50)
            assert (int(ref["slice_begin"][t]), int(ref["slice_len"][t]), int(ref["score_final"][t]), ref["btr"][t]) == \
                   (want["slice_begin"], want["slice_len"], want["score_final"], want["btr"]), t
    finally:
        c.close()
