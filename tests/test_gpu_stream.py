"""The stream-ordered pipelines (tracy_amd/csrc/stream.hip: planned on the device, one host synchronisation per call) against the
pipelines planned by the host between launches (pipeline.hip, option no_stream) and against the oracle: same arrays, bit for bit --
on the batches the stream-ordered pass is built for (nothing falls back) AND on traces whose certificates fail or whose bands are
too wide (the dead traces are re-done by the host-planned tiers and scattered back)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SC = (3, -5, -10, -4)
ALIGN_KEYS = ("forward", "score_fwd", "score_rev", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final")


@pytest.fixture(scope="module")
def ctx():
    import tracy_amd
    c = tracy_amd.Context(0)
    yield c
    c.close()


def both_ways(ctx, fn):
    """fn() on the stream-ordered pipeline and on the host-planned one; (stream result, its stats, host-planned result, its stats)"""
    ctx.set_option("no_stream", 0)
    a = fn()
    sa = ctx.last_call_stats()
    ctx.set_option("no_stream", 1)
    b = fn()
    sb = ctx.last_call_stats()
    ctx.set_option("no_stream", 0)
    return a, sa, b, sb


def same_align(a, b, exact, what=""):
    keys = ALIGN_KEYS if exact else tuple(k for k in ALIGN_KEYS if k not in ("score_fwd", "score_rev"))
    for k in keys:
        assert np.array_equal(a[k], b[k]), (what, k, np.nonzero(np.asarray(a[k]) != np.asarray(b[k]))[0][:8])
    assert a["btr"] == b["btr"], what


@pytest.mark.parametrize("exact", [True, False])
def test_align_uniform_batch_is_stream_ordered(ctx, exact):
    """the bench shape in miniature: every trace certifies on the device, one synchronisation, nothing falls back"""
    from tracy_amd import hostlib
    refs, profs, rev = hostlib.synth_align(31, 96, 4000, 1000, 2)
    refl = [r.tobytes() for r in refs]
    a, sa, b, sb = both_ways(ctx, lambda: ctx.align_traces(list(profs), refl, SC, 50, 50, exact_scores=exact))
    assert sa["stream_ordered"] == 1 and sb["stream_ordered"] == 0, (sa, sb)
    assert sa["fallback_traces"] == 0, sa
    assert sa["host_syncs"] <= 2 and sb["host_syncs"] >= 6, (sa, sb)  # (host buffers: one more for the copy-out)
    assert sa["pruned"] == 96 and sa["final_banded"] == 96, sa
    same_align(a, b, exact)
    assert [int(x) for x in a["forward"]] == [1 - int(r) for r in rev]


@pytest.mark.parametrize("exact,quads", [(True, False), (False, False), (True, True)])
def test_align_failing_certificates_fall_back_per_trace(ctx, exact, quads):
    """test_gpu_front's cases (the target twice, chimeras, long indels, cut windows, mixed strip heights): the traces the device cannot
    certify are re-done by the host-planned tiers; every array equals the host-planned pipeline's and the oracle's.  quads: with the
    narrow first tier and the later tiers over device-side lists (large batches' form), where some units pass all three uncertified"""
    import sage_oracle as so
    from test_gpu_front import cases
    rng = np.random.default_rng(77)
    cs = cases(rng)
    profs, wins = [c[0] for c in cs], [c[1] for c in cs]
    if quads:
        ctx.set_option("quad_tier_min", 0)
        ctx.set_option("front_list_min", 0)
    try:
        a, sa, b, sb = both_ways(ctx, lambda: ctx.align_traces(profs, wins, SC, 50, 50, exact_scores=exact))
    finally:
        ctx.set_option("quad_tier_min", 32768)
        ctx.set_option("front_list_min", 1024)
    assert sa["stream_ordered"] == 1, sa
    assert 4 <= sa["fallback_traces"] <= len(cs) - 16, sa  # both outcomes are exercised
    same_align(a, b, exact, "stream vs host-planned")
    for i in range(0, len(cs), 3):
        w = so.align_trace(profs[i], wins[i], SC, 50, 50)
        for k in ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
            assert int(a[k][i]) == int(w[k]), (i, cs[i][2], k)
        assert a["btr"][i] == w["btr"], (i, cs[i][2])


def test_two_different_batches_back_to_back_with_tier_lists(ctx):
    """with lists on, a tier visits only the units the tier before left: the slots of the others must not keep a verdict of the call
    before.  Two different batches, one after the other on one context, lists and the narrow tier forced; each equals the host-planned
    pipeline's result and a fresh context's"""
    import tracy_amd
    from tracy_amd import hostlib
    from test_gpu_front import cases
    rng = np.random.default_rng(77)
    cs = cases(rng)
    batch_a = ([c[0] for c in cs], [c[1] for c in cs])                    # certificates of every kind fail for some units
    refs, profs, rev = hostlib.synth_align(131, len(cs), 3500, 900, 2)  # every unit certifies in the first tier
    batch_b = (list(profs), [r.tobytes() for r in refs])
    ctx.set_option("quad_tier_min", 0)
    ctx.set_option("front_list_min", 1)
    try:
        got = [ctx.align_traces(p_, w_, SC, 50, 50) for p_, w_ in (batch_a, batch_b, batch_a)]
    finally:
        ctx.set_option("quad_tier_min", 32768)
        ctx.set_option("front_list_min", 1024)
    ctx.set_option("no_stream", 1)
    want = [ctx.align_traces(p_, w_, SC, 50, 50) for p_, w_ in (batch_a, batch_b)]
    ctx.set_option("no_stream", 0)
    same_align(got[0], want[0], True, "first batch")
    same_align(got[1], want[1], True, "second batch, after a different one")
    same_align(got[2], want[0], True, "first batch again")
    fresh = tracy_amd.Context(0)
    fresh.set_option("quad_tier_min", 0)
    fresh.set_option("front_list_min", 1)
    same_align(fresh.align_traces(batch_b[0], batch_b[1], SC, 50, 50), got[1], True, "fresh context")
    fresh.close()


def test_align_stream_with_device_buffers_and_lanes(ctx):
    """device-resident inputs and results (what bench.py times), one and two lanes"""
    from tracy_amd import hostlib
    nt, n, mf = 256, 3000, 900
    refs, profs, rev = hostlib.synth_align(5, nt, n, mf, 2)
    refl = [r.tobytes() for r in refs]
    want = ctx.align_traces(list(profs), refl, SC, 50, 50)
    for lanes in (1, 2):
        ctx.set_lanes(lanes)
        got = ctx.align_traces(list(profs), refl, SC, 50, 50, device=True)
        st = ctx.last_call_stats()
        assert st["stream_ordered"] == 1 and st["fallback_traces"] == 0, st
        same_align(got, want, True, "lanes %d" % lanes)
    ctx.set_lanes(1)


def test_options_are_read_once_and_described(ctx, monkeypatch):
    """the environment is read when a context is created; afterwards only tracyhip_set_option changes a switch"""
    import tracy_amd
    monkeypatch.setenv("TRACYHIP_NO_FRONT", "1")
    c2 = tracy_amd.Context(0)
    monkeypatch.delenv("TRACYHIP_NO_FRONT")
    try:
        assert c2.describe()["no_front"] == "1" and ctx.describe()["no_front"] == "0"
        monkeypatch.setenv("TRACYHIP_NO_BAND16", "1")  # too late for both
        assert c2.describe()["no_band16"] == "0" and ctx.describe()["no_band16"] == "0"
        monkeypatch.delenv("TRACYHIP_NO_BAND16")
        c2.set_option("NO_FRONT", 0)
        c2.set_option("band_w", 12)
        d = c2.describe()
        assert d["no_front"] == "0" and d["band_w"] == "12"
        with pytest.raises(tracy_amd.capi.TracyHipError):
            c2.set_option("no_such_switch", 1)
    finally:
        c2.close()


# ---- tracy decompose ---------------------------------------------------------------------------------------------------
def same_decompose(a, b, what=""):
    for k in a:
        x, y = a[k], b[k]
        if k in ("dcp_indel", "dcp_err", "ops"):  # raw tables: compared through "dcp" (rows written) and "btr*"
            continue
        if k == "bp":
            assert [(v.indelshift, v.traceleft, v.breakpoint, v.best_diff) for v in x] == [(v.indelshift, v.traceleft, v.breakpoint, v.best_diff) for v in y], (what, k)
        elif k == "dstatus":
            assert [(v.kind, v.best_ins, v.best_del, v.best_fr, v.dcp_n) for v in x] == [(v.kind, v.best_ins, v.best_del, v.best_fr, v.dcp_n) for v in y], (what, k)
        elif isinstance(x, np.ndarray):
            assert np.array_equal(x, y), (what, k, np.nonzero(x != y)[0][:8])
        else:
            assert x == y, (what, k)


def decompose_both_ways(ctx, d, refs, nd, exact=True, **kw):
    from tracy_amd import capi

    def run():
        hbc = capi.HostBaseCalls([d["signal"][i] for i in range(nd)], [d["bcpos"][i] for i in range(nd)],
                                 [d["primary"][i].tobytes() for i in range(nd)], [d["secondary"][i].tobytes() for i in range(nd)])
        return ctx.decompose_traces([d["profiles"][i] for i in range(nd)], hbc, refs, SC, exact_scores=exact, **kw)
    return both_ways(ctx, run)


@pytest.mark.parametrize("exact", [True, False])
def test_decompose_uniform_batch_is_stream_ordered(ctx, exact):
    """configs[2] in miniature (het indels + SNVs, homozygous indels, no variant; both strands): one synchronisation per call (+ the
    copy-out of host buffers), every array the host-planned pipeline's"""
    from tracy_amd import hostlib
    nd = 160
    d = hostlib.synth_decompose_batch(4711, nd, 3000, 1000, 0, mix=1)
    refs = [d["refs"][i].tobytes() for i in range(nd)]
    a, sa, b, sb = decompose_both_ways(ctx, d, refs, nd, exact)
    assert sa["stream_ordered"] == 1 and sb["stream_ordered"] == 0, (sa, sb)
    assert sa["fallback_traces"] <= nd // 20, sa
    assert sa["host_syncs"] <= 2 + (2 if sa["fallback_traces"] else 0) + 20 * (sa["fallback_traces"] > 0) and sb["host_syncs"] >= 20, (sa, sb)
    assert sa["allele_banded"][0] >= nd - nd // 20 and sa["allele_banded"][2] >= nd - nd // 20, sa
    same_decompose(a, b, "exact %s" % exact)
    assert int((np.asarray(a["status"]) == 0).sum()) > nd // 2


@pytest.mark.parametrize("quads", [False, True])
def test_decompose_failing_certificates_fall_back_per_trace(ctx, quads):
    """windows that hold the locus twice cannot certify the pruned sweeps (test_gpu_decompose): those traces are re-done by the
    host-planned tiers from their untouched basecalls; the rest stays stream-ordered; everything equals the host-planned run and the
    oracle.  quads: the tiers of a large batch (narrow tier first, later tiers over device-side lists of what is left)"""
    from indigo_oracle import decompose_trace
    from tracy_amd import hostlib
    nd = 72
    d = hostlib.synth_decompose_batch(919, nd, 2200, 640, 0, mix=1)
    rng = np.random.default_rng(5)
    refs = []
    for i in range(nd):
        r = d["refs"][i].tobytes()
        if i % 3 == 0:
            c = bytearray(r)
            for j in rng.integers(0, len(c), 12):
                c[int(j)] = int(rng.choice(list(b"ACGT")))
            r = r + bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(40, 300))).tolist()) + bytes(c)
        refs.append(r)
    if quads:
        ctx.set_option("quad_tier_min", 0)
        ctx.set_option("front_list_min", 0)
    try:
        a, sa, b, sb = decompose_both_ways(ctx, d, refs, nd)
    finally:
        ctx.set_option("quad_tier_min", 32768)
        ctx.set_option("front_list_min", 1024)
    assert sa["stream_ordered"] == 1 and 8 <= sa["fallback_traces"] <= nd - 16, sa
    same_decompose(a, b)
    for i in (0, 1, 3, 10, 30):
        want = decompose_trace(d["signal"][i], d["bcpos"][i], d["primary"][i].tobytes(), d["secondary"][i].tobytes(), refs[i], SC)
        if want["status"] != 0:
            continue
        assert a["btr0"][i] == want["btr0"] and a["btr1"][i] == want["btr1"] and a["btr2"][i] == want["btr2"], i
        assert a["primary"][i] == want["primary"] and a["secondary"][i] == want["secondary"] and a["dcp"][i] == want["dcp"], i


def test_decompose_ragged_batch_and_lanes(ctx):
    """traces of several lengths (strip heights 8 .. 16) in one call, one and two lanes"""
    from tracy_amd import capi, hostlib
    parts = [hostlib.synth_decompose_batch(100 + i, 48, 1800 + 400 * i, mf, 0, mix=1) for i, mf in enumerate((420, 640, 900, 1010))]
    d = {k: sum((list(p[k]) for p in parts), []) for k in ("signal", "bcpos", "primary", "secondary", "profiles", "refs")}
    nd = len(d["profiles"])
    refs = [r.tobytes() for r in d["refs"]]
    a, sa, b, sb = decompose_both_ways(ctx, d, refs, nd)
    assert sa["stream_ordered"] == 1 and sa["fallback_traces"] <= nd // 8, sa
    same_decompose(a, b, "ragged")
    ctx.set_lanes(2)
    try:
        c, sc_, _, _ = decompose_both_ways(ctx, d, refs, nd)
        assert sc_["stream_ordered"] == 1, sc_
        same_decompose(c, b, "ragged, two lanes")
    finally:
        ctx.set_lanes(1)


def test_quad_tier_of_the_pruned_sweeps_changes_nothing(ctx):
    """the narrow first tier of the pruned sweeps (c* +- 5, four lanes per pair; batches of 32 768 units and more by default) forced
    on a small batch: what it certifies it certifies with the score and end the wide tiers find; every array equals the run without it"""
    from tracy_amd import capi, hostlib
    refs, profs, rev = hostlib.synth_align(78, 128, 3500, 950, 2)
    refl = [r.tobytes() for r in refs]
    nd = 128
    d = hostlib.synth_decompose_batch(616, nd, 2600, 900, 0, mix=1)
    drefs = [d["refs"][i].tobytes() for i in range(nd)]

    def run():
        a = ctx.align_traces(list(profs), refl, SC, 50, 50)
        sa = ctx.last_call_stats()
        hbc = capi.HostBaseCalls([d["signal"][i] for i in range(nd)], [d["bcpos"][i] for i in range(nd)],
                                 [d["primary"][i].tobytes() for i in range(nd)], [d["secondary"][i].tobytes() for i in range(nd)])
        return a, sa, ctx.decompose_traces([d["profiles"][i] for i in range(nd)], hbc, drefs, SC), ctx.last_call_stats()
    a0, sa0, b0, sb0 = run()
    ctx.set_option("quad_tier_min", 0)
    ctx.set_option("front_list_min", 0)
    try:
        assert ctx.describe()["quad_tier_min"] == "0" and ctx.describe()["front_list_min"] == "0"
        a1, sa1, b1, sb1 = run()
        # lists without the quad tier: the last tier's only
        ctx.set_option("quad_tier_min", 32768)
        a3, sa3, b3, sb3 = run()
        ctx.set_option("quad_tier_min", 0)
        # the later tiers over device-side lists of what is left (the default with the quad tier) against skipping in place
        ctx.set_option("no_front_lists", 1)
        assert ctx.describe()["no_front_lists"] == "1"
        a2, sa2, b2, sb2 = run()
    finally:
        ctx.set_option("quad_tier_min", 32768)
        ctx.set_option("front_list_min", 1024)
        ctx.set_option("no_front_lists", 0)
    same_align(a3, a0, True, "lists, no quad tier")
    same_decompose(b3, b0, "lists, no quad tier")
    assert sb3["allele_shared_prefix"] == sb1["allele_shared_prefix"], (sb1, sb3)
    assert sa1["stream_ordered"] == 1 and sb1["stream_ordered"] == 1
    assert sa1["pruned"] == sa0["pruned"] and sb1["allele_pruned"] == sb0["allele_pruned"], (sa0, sa1, sb0, sb1)
    # (with the lists, the second allele of a trace reads the first one's kept prefix row where both begin with the same 128 characters)
    assert sb1["allele_shared_prefix"] >= nd // 4 and sb2["allele_shared_prefix"] == 0 and sb0["allele_shared_prefix"] == 0, (sb0, sb1, sb2)
    sb1 = dict(sb1, allele_shared_prefix=0)
    assert sa2 == sa1 and sb2 == sb1, (sa1, sa2, sb1, sb2)
    same_align(a1, a0, True, "quad tier")
    same_decompose(b1, b0, "quad tier")
    same_align(a2, a0, True, "quad tier, tiers in place")
    same_decompose(b2, b0, "quad tier, tiers in place")


@pytest.mark.parametrize("option", ["no_quads", "no_fork", "no_origin_band"])
def test_quad_form_and_side_streams_change_nothing(ctx, option):
    """narrow bands four lanes to a pair (band16.h, P = 4) and the stages that run side by side on the context's side streams (the two
    strands of the orientation stage, the lists of a band stage, allelicFraction beside the allele stages) are ways of running the same
    kernels over the same bands: every array equals the run without them"""
    from tracy_amd import hostlib
    refs, profs, rev = hostlib.synth_align(77, 160, 3500, 950, 2)
    refl = [r.tobytes() for r in refs]
    nd = 144
    d = hostlib.synth_decompose_batch(515, nd, 2600, 900, 0, mix=1)
    drefs = [d["refs"][i].tobytes() for i in range(nd)]

    def run():
        from tracy_amd import capi
        a = ctx.align_traces(list(profs), refl, SC, 50, 50)
        sa = ctx.last_call_stats()
        hbc = capi.HostBaseCalls([d["signal"][i] for i in range(nd)], [d["bcpos"][i] for i in range(nd)],
                                 [d["primary"][i].tobytes() for i in range(nd)], [d["secondary"][i].tobytes() for i in range(nd)])
        b = ctx.decompose_traces([d["profiles"][i] for i in range(nd)], hbc, drefs, SC)
        return a, sa, b, ctx.last_call_stats()
    a1, sa1, b1, sb1 = run()
    ctx.set_option(option, 1)
    try:
        assert ctx.describe()[option] == "1"
        a0, sa0, b0, sb0 = run()
    finally:
        ctx.set_option(option, 0)
    assert sa1["stream_ordered"] == 1 and sb1["stream_ordered"] == 1 and sa0["stream_ordered"] == 1 and sb0["stream_ordered"] == 1
    assert sa1["final_banded"] == sa0["final_banded"] and sb1["allele_banded"] == sb0["allele_banded"]
    same_align(a1, a0, True, option)
    same_decompose(b1, b0, option)


@pytest.mark.parametrize("seed", [9001, 9002, 9003])
def test_decompose_low_complexity_windows_walk_the_reference_path(ctx, seed):
    """windows of short tandem repeats: gaps that can sit anywhere in a run, alignments that can start a repeat unit earlier -- co-optimal
    paths everywhere.  gotoh(allele, slice) runs on the g + 3 diagonals that hold the path the origin sweep followed (the one the
    reference's traceback walks), not on the 2 g + 3 of every co-optimal path: the strings must be the oracle's (indigo.h:355-387),
    the host-planned pipeline's, and those of the wide band (option no_origin_band)"""
    from indigo_oracle import decompose_trace
    from tracy_amd import hostlib
    nd = 96
    d = hostlib.synth_decompose_batch(seed, nd, 2400, 800, 0, mix=2)
    refs = [d["refs"][i].tobytes() for i in range(nd)]
    a, sa, b, sb = decompose_both_ways(ctx, d, refs, nd)
    assert sa["stream_ordered"] == 1, sa
    ctx.set_option("no_origin_band", 1)
    try:
        c, sc_, _, _ = decompose_both_ways(ctx, d, refs, nd)
    finally:
        ctx.set_option("no_origin_band", 0)
    same_decompose(a, b, "stream vs host-planned")
    same_decompose(a, c, "path band vs every co-optimal path's band")
    # enough traces must have gone through the band stages on the device for the comparison to mean something
    assert sa["allele_banded"][0] - sa["fallback_traces"] >= nd // 3, sa
    checked = 0
    for i in range(nd):
        want = decompose_trace(d["signal"][i], d["bcpos"][i], d["primary"][i].tobytes(), d["secondary"][i].tobytes(), refs[i], SC)
        assert int(a["status"][i]) == want["status"], i
        if want["status"] != 0:
            continue
        checked += 1
        assert a["btr0"][i] == want["btr0"] and a["btr1"][i] == want["btr1"] and a["btr2"][i] == want["btr2"], i
        assert a["primary"][i] == want["primary"] and a["secondary"][i] == want["secondary"], i
    assert checked >= nd // 4, checked


def test_batch_planned_on_the_host_worker_threads(ctx):
    """from 16 384 traces on the per-trace planning loops of a call run on the library's worker threads (parallel_for): a batch of that
    size, ragged so that the sweeps are ordered by strip height and size, equals the same traces aligned in small batches"""
    from tracy_amd import hostlib
    nt = 16384 + 300
    refs, profs, rev = hostlib.synth_align(4100, 512, 900, 330, 2)
    rng = np.random.default_rng(5)
    pick = rng.integers(0, 512, size=nt)
    cut = rng.integers(0, 60, size=nt)  # ragged lengths
    plist = [np.ascontiguousarray(profs[pick[i]][:, :330 - int(cut[i])]) for i in range(nt)]
    rlist = [refs[pick[i]].tobytes()[:900 - int(cut[i]) // 2] for i in range(nt)]
    got = ctx.align_traces(plist, rlist, SC, 20, 20)
    st = ctx.last_call_stats()
    assert st["stream_ordered"] == 1, st
    for lo in (0, 8000, nt - 128):
        want = ctx.align_traces(plist[lo:lo + 128], rlist[lo:lo + 128], SC, 20, 20)
        for k in ALIGN_KEYS:
            assert np.array_equal(np.asarray(got[k][lo:lo + 128]), np.asarray(want[k])), (lo, k)
        assert got["btr"][lo:lo + 128] == want["btr"], lo
