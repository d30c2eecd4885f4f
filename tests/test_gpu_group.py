"""Device groups (tracyhip_group_*: the GPUs of a node behind one handle, one host thread per member) and the asynchronous
entry points, on the one GPU the test box has: a group may list a device twice, which exercises the same block cutting,
threads and result placement as eight devices would."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SC = (3, -5, -10, -4)


@pytest.fixture(scope="module")
def ctx():
    import tracy_amd
    c = tracy_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def group():
    import tracy_amd
    g = tracy_amd.Group([0, 0, 0])
    assert g.size() == 3
    yield g
    g.close()


def same(a, b):
    for k in a:
        if k in ("ops", "dcp_indel", "dcp_err"):  # raw buffers: bytes past ops_len[i] / rows past dcp_n are unspecified (staging
            continue                               # buffers are reused); "btr" and "dcp" hold what was written
        x, y = a[k], b[k]
        if isinstance(x, np.ndarray):
            assert np.array_equal(x, y), k
        elif isinstance(x, (list, tuple, dict, int, float, str, bytes)):
            assert x == y, k
        elif k == "bp":
            assert [(v.indelshift, v.traceleft, v.breakpoint, v.best_diff) for v in x] == [(v.indelshift, v.traceleft, v.breakpoint, v.best_diff) for v in y]
        else:
            assert bytes(x) == bytes(y), k


def test_group_align_traces(ctx, group):
    from tracy_amd import hostlib
    refs, profs, rev = hostlib.synth_align(31, 230, 2500, 700, 0)
    refl = [r.tobytes() for r in refs]
    for exact in (True, False):
        one = ctx.align_traces(list(profs), refl, SC, 50, 50, exact_scores=exact)
        many = group.align_traces(list(profs), refl, SC, 50, 50, exact_scores=exact)
        if not exact:  # the loser's score may be its bound in either run
            for r in (one, many):
                r.pop("score_fwd"), r.pop("score_rev")
        same(one, many)
    group.set_lanes(2)  # members may split their block further
    try:
        same(ctx.align_traces(list(profs), refl, SC, 50, 50), group.align_traces(list(profs), refl, SC, 50, 50))
    finally:
        group.set_lanes(1)


def test_group_decompose_traces(ctx, group):
    from tracy_amd import capi, hostlib
    nd = 200
    d = hostlib.synth_decompose_batch(99, nd, 1500, 520, 0, mix=1)

    def run(c):
        hbc = capi.HostBaseCalls([d["signal"][i] for i in range(nd)], [d["bcpos"][i] for i in range(nd)],
                                 [d["primary"][i].tobytes() for i in range(nd)], [d["secondary"][i].tobytes() for i in range(nd)])
        return c.decompose_traces([d["profiles"][i] for i in range(nd)], hbc, [d["refs"][i].tobytes() for i in range(nd)], SC)
    same(run(ctx), run(group))


def test_group_all_pairs_and_msa(ctx, group):
    from tracy_amd import hostlib, msalib
    from tracy_amd.shard import pair_bounds
    refs, profs, rev = hostlib.synth_align(5, 40, 900, 300, 0)
    plist = [np.ascontiguousarray(profs[i][:, :int(200 + 2 * i)]) for i in range(40)]  # ragged
    i1, i2 = msalib.pair_list(40)
    one = ctx.score(plist, plist, SC + (1, 1), idx1=i1, idx2=i2)
    many = group.score(plist, plist, SC + (1, 1), idx1=i1, idx2=i2)
    assert np.array_equal(one, many)
    b = pair_bounds([p.shape[1] for p in plist], 3)
    assert b[0] == 0 and b[-1] == len(i1) and 0 < b[1] < b[2] < len(i1)
    few = plist[:12]
    assert msalib.msa(ctx, few, SC) == msalib.msa(ctx, few, SC, group=group)


def test_async_calls_on_two_contexts(ctx):
    """tracyhip_align_traces_async returns at once; two contexts work concurrently; results are final after synchronize()"""
    import tracy_amd
    from tracy_amd import capi, hostlib
    refs, profs, rev = hostlib.synth_align(77, 96, 3000, 800, 0)
    refl = [r.tobytes() for r in refs]
    want = [ctx.align_traces(list(profs[lo:lo + 48]), refl[lo:lo + 48], SC, 50, 50) for lo in (0, 48)]
    c2 = tracy_amd.Context(0)
    try:
        preps = [capi.PreparedAlign(list(profs[lo:lo + 48]), refl[lo:lo + 48], SC, 50, 50) for lo in (0, 48)]
        for c, p in zip((ctx, c2), preps):
            c.align_traces_async(p.job, p.prm, p.out, capi.MEM_HOST)
        # queue a second, different call behind the first one on the same context: issue order is execution order
        again = capi.PreparedAlign(list(profs[:48]), refl[:48], SC, 0, 0)
        ctx.align_traces_async(again.job, again.prm, again.out, capi.MEM_HOST)
        ctx.synchronize()
        c2.synchronize()
        for p, w in zip(preps, want):
            same(w, p.results())
        same(ctx.align_traces(list(profs[:48]), refl[:48], SC, 0, 0), again.results())
        # an error inside an asynchronous call surfaces at synchronize()
        bad = capi.PreparedAlign(list(profs[:4]), refl[:4], (3, -5, -10, -4), 50, 50)
        bad.prm.match = 100000
        ctx.align_traces_async(bad.job, bad.prm, bad.out, capi.MEM_HOST)
        with pytest.raises(capi.TracyHipError) as ei:
            ctx.synchronize()
        assert ei.value.code == capi.ERR_RANGE
        ctx.synchronize()  # reported once
    finally:
        c2.close()
