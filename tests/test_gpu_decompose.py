"""GPU parity of the allele-deconvolution kernels (decompose.h) through the C ABI, against the oracle."""
import numpy as np
import pytest

import pyoracle as orc
from decomp_cases import SC, case_list, oracle_decompose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import tracy_amd
    c = tracy_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def cases():
    return case_list()


def test_find_breakpoint(ctx, cases):
    rng = np.random.default_rng(0)
    profs = [c["prof"] for c in cases]
    flat = np.zeros((6, 300), dtype=np.float32)
    flat[:4] = 0.25
    short = np.ascontiguousarray(cases[0]["prof"][:, :20])
    noise = np.zeros((6, 777), dtype=np.float32)
    x = rng.random((4, 777)).astype(np.float32)
    noise[:4] = x / x.sum(axis=0)
    profs += [flat, short, noise]
    got = ctx.find_breakpoint(profs)
    for p, g in zip(profs, got):
        w = orc.find_breakpoint(p)
        assert (g.indelshift, g.traceleft, g.breakpoint) == (w.indelshift, w.traceleft, w.breakpoint)
        assert np.float32(g.best_diff) == np.float32(w.bestDiff)


def test_homozygous_breakpoint(ctx, cases):
    from tracy_amd import capi
    rows = [c["rows"] for c in cases] + [(b"----", b"ACGT"), (b"ACGTACGTAC" * 3, b"ACGTACGTAC" * 3)]
    bps = [capi.Breakpoint(0, 1, 0, 0.0) for _ in rows]
    got, status = ctx.find_homozygous_breakpoint(rows, bps)
    for r, g, s in zip(rows, got, status):
        rc, w = orc.find_homozygous_breakpoint(r[0], r[1])
        assert int(s) == rc
        if rc == 1:
            assert (g.indelshift, g.traceleft, g.breakpoint) == (w.indelshift, w.traceleft, w.breakpoint)
            assert np.float32(g.best_diff) == np.float32(w.bestDiff)
    # traces with an indel shift are left untouched (indigo.h:314-317)
    bps = [capi.Breakpoint(1, 0, 123, 0.5) for _ in rows]
    got, status = ctx.find_homozygous_breakpoint(rows, bps)
    assert all(g.breakpoint == 123 and g.indelshift == 1 for g in got)


def test_homozygous_breakpoint_random_rows(ctx):
    from tracy_amd import capi
    from decomp_cases import random_row_pairs
    rows = random_row_pairs(77, 400)
    bps = [capi.Breakpoint(0, 1, 0, 0.0) for _ in rows]
    got, status = ctx.find_homozygous_breakpoint(rows, bps)
    seen = set()
    for r, g, s in zip(rows, got, status):
        rc, w = orc.find_homozygous_breakpoint(r[0], r[1])
        assert int(s) == rc
        seen.add((rc, w.indelshift if rc == 1 else 0))
        if rc == 1:
            assert (g.indelshift, g.traceleft, g.breakpoint) == (w.indelshift, w.traceleft, w.breakpoint)
            assert np.float32(g.best_diff) == np.float32(w.bestDiff)
    assert seen == {(0, 0), (-1, 0), (1, 0), (1, 1)}


def test_decompose_chain(ctx, cases):
    """decomposeAlleles -> generateSecondaryDecomposed -> allelicFraction, batched, vs the oracle"""
    from tracy_amd import capi
    want = [oracle_decompose(c) for c in cases]
    hbc = capi.HostBaseCalls([c["sig"] for c in cases], [c["bcpos"] for c in cases], [c["pri"] for c in cases],
                             [c["sec"] for c in cases])
    bps = [capi.Breakpoint(c["bp"].indelshift, c["bp"].traceleft, c["bp"].breakpoint, c["bp"].bestDiff) for c in cases]
    pri, sec, dcp, status = ctx.decompose_alleles(hbc, [c["rows"] for c in cases], bps, [len(c["ref"]) for c in cases])
    for i, w in enumerate(want):
        assert pri[i] == w["pri"] and sec[i] == w["sec"], i
        assert dcp[i] == w["dcp"], i
        assert status[i][0] == w["status"][0]
        if w["status"][0] != 0:
            assert status[i] == w["status"]
    sd = ctx.secondary_decomposed(hbc)
    for i, w in enumerate(want):
        assert hbc.split(sd)[i] == w["secdecomp"]
    fr = ctx.allelic_fraction(hbc, sd, 50, 50)
    for i, w in enumerate(want):
        assert (float(fr[i, 0]), float(fr[i, 1])) == w["af"], (i, fr[i], w["af"])
    assert any(w["af"] != (0.5, 0.5) for w in want)


def test_decompose_parameter_edges(ctx, cases):
    from tracy_amd import capi
    c = cases[0]
    for (mi, madc) in [(1, 5), (7, 5), (40, 0), (300, 9)]:
        hbc = capi.HostBaseCalls([c["sig"]], [c["bcpos"]], [c["pri"]], [c["sec"]])
        bps = [capi.Breakpoint(c["bp"].indelshift, c["bp"].traceleft, c["bp"].breakpoint, c["bp"].bestDiff)]
        pri, sec, dcp, status = ctx.decompose_alleles(hbc, [c["rows"]], bps, [len(c["ref"])], 50, 50, mi, madc)
        w = orc.decompose_alleles(c["rows"][0], c["rows"][1], c["pri"], c["sec"], c["bp"], len(c["ref"]), 50, 50, mi, madc)
        assert (pri[0], sec[0], dcp[0]) == (w[0], w[1], w[2]) and status[0][0] == w[3][0]
    # allelic fraction: identical alleles -> (0.5, 0.5); untrimmed corner
    hbc = capi.HostBaseCalls([c["sig"]], [c["bcpos"]], [c["pri"]], [c["pri"]])
    fr = ctx.allelic_fraction(hbc, np.frombuffer(c["pri"], dtype=np.uint8), 50, 50)
    assert tuple(fr[0]) == (0.5, 0.5)


def test_decompose_traces_pipeline(ctx):
    """tracyhip_decompose_traces == indigo.h:190-388 composed from the oracle"""
    from indigo_oracle import decompose_trace
    from sage_oracle import revcomp
    from tracy_amd import capi, hostlib
    sigs, poss, refs, bcs = [], [], [], []
    for i, (seed, kind, frac) in enumerate([(21, 0, 0.6), (22, 0, 0.55), (23, 1, 0.6), (24, 0, 0.7), (25, 0, 0.5)]):
        ref, sig, pos, indel = hostlib.synth_decompose(seed, 1400, 480, 25, kind, frac)
        if i % 2:  # the trace reads the reverse strand of its reference window
            ref = revcomp(ref)
        pri, sec, con, bcpos = hostlib.basecall(sig, pos, 0.33)
        sigs.append(sig); poss.append(pos); refs.append(ref); bcs.append((pri, sec, bcpos))
    profs = [hostlib.create_profile(sigs[i], bcs[i][2], bcs[i][0], bcs[i][1], 0, 0) for i in range(len(sigs))]
    hbc = capi.HostBaseCalls(sigs, [b[2] for b in bcs], [b[0] for b in bcs], [b[1] for b in bcs])
    got = ctx.decompose_traces(profs, hbc, refs, SC)
    for i in range(len(sigs)):
        w = decompose_trace(sigs[i], bcs[i][2], bcs[i][0], bcs[i][1], refs[i], SC)
        assert int(got["status"][i]) == w["status"] == 0
        for k in ("score_fwd", "score_rev", "forward", "score_trim"):
            assert int(got[k][i]) == int(w[k]), (i, k)
        g = got["bp"][i]
        assert (g.indelshift, g.traceleft, g.breakpoint) == (w["bp"].indelshift, w["bp"].traceleft, w["bp"].breakpoint)
        assert got["primary"][i] == w["primary"] and got["secondary"][i] == w["secondary"]
        assert got["secdecomp_list"][i] == w["secdecomp"]
        assert got["dcp"][i] == w["dcp"]
        assert (float(got["fractions"][2 * i]), float(got["fractions"][2 * i + 1])) == w["af"]
        for k in range(3):
            assert int(got["score%d" % k][i]) == w["score%d" % k], (i, k)
            assert got["btr%d" % k][i] == w["btr%d" % k], (i, k)
        for k in range(2):
            for nm in ("slice_begin", "slice_len", "ref_pos"):
                assert int(got["%s%d" % (nm, k)][i]) == int(w["%s%d" % (nm, k)]), (i, nm, k)
    assert sorted(got["forward"].tolist()) == [0, 0, 1, 1, 1]


def test_decompose_traces_lanes(ctx):
    """tracyhip_set_lanes: tracyhip_decompose_traces split over chunks in flight returns what the single-lane call returns
    (host-staged buffers: every payload and result region travels with its chunk)"""
    import tracy_amd
    from tracy_amd import capi, hostlib
    nd = 260
    d = hostlib.synth_decompose_batch(4242, nd, 1500, 520, 0)

    def run(c):
        hbc = capi.HostBaseCalls([d["signal"][i] for i in range(nd)], [d["bcpos"][i] for i in range(nd)],
                                 [d["primary"][i].tobytes() for i in range(nd)], [d["secondary"][i].tobytes() for i in range(nd)])
        return c.decompose_traces([d["profiles"][i] for i in range(nd)], hbc, [d["refs"][i].tobytes() for i in range(nd)], SC)
    one = run(ctx)
    c3 = tracy_amd.Context(0)
    c3.set_lanes(3)
    try:
        many = run(c3)
    finally:
        c3.set_lanes(1)
    assert int((np.asarray(one["status"]) == 0).sum()) > nd // 2
    for k in one:
        a, b = one[k], many[k]
        if k in ("dcp_indel", "dcp_err"):  # raw tables: rows past dstatus[t].dcp_n are unspecified ("dcp" holds the rows written)
            continue
        if k == "bp":
            assert [(x.indelshift, x.traceleft, x.breakpoint, x.best_diff) for x in a] == [(x.indelshift, x.traceleft, x.breakpoint, x.best_diff) for x in b]
        elif isinstance(a, np.ndarray):
            assert np.array_equal(a, b), k
        elif isinstance(a, (list, tuple, dict, int, float, str, bytes)):
            assert a == b, k
        else:  # ctypes arrays of result records
            assert bytes(a) == bytes(b), k


def test_origin_sweep_on_the_certified_sub_window(ctx, monkeypatch):
    """gotoh(allele, window) is only read by trimReferenceSlice; the origin-tracking sweep runs on the columns a 16-bit score
    sweep certifies (score and row-m end bound where an optimal path can start).  Same results as on the whole window, on
    the configs[2] mix and on windows whose alleles sit at the very beginning / end"""
    from tracy_amd import capi, hostlib
    nd = 160
    d = hostlib.synth_decompose_batch(606, nd, 3000, 800, 0, mix=1)

    def run():
        hbc = capi.HostBaseCalls([d["signal"][i] for i in range(nd)], [d["bcpos"][i] for i in range(nd)],
                                 [d["primary"][i].tobytes() for i in range(nd)], [d["secondary"][i].tobytes() for i in range(nd)])
        return ctx.decompose_traces([d["profiles"][i] for i in range(nd)], hbc, [d["refs"][i].tobytes() for i in range(nd)], SC)
    cut = run()
    ctx.set_option("no_subwindow", 1)
    whole = run()
    ctx.set_option("no_subwindow", 0)
    assert int((np.asarray(cut["status"]) == 0).sum()) > nd // 2
    for k in cut:
        a, b = cut[k], whole[k]
        if k in ("dcp_indel", "dcp_err", "ops"):
            continue
        if k == "bp":
            assert [(x.indelshift, x.traceleft, x.breakpoint, x.best_diff) for x in a] == [(x.indelshift, x.traceleft, x.breakpoint, x.best_diff) for x in b]
        elif isinstance(a, np.ndarray):
            assert np.array_equal(a, b), k
        elif isinstance(a, (list, tuple, dict, int, float, str, bytes)):
            assert a == b, k
        else:
            assert bytes(a) == bytes(b), k


def test_decompose_beyond_the_default_size_class(ctx):
    """maxindel > 1024 and traces of >= 2048 basecalls: the scan state of decomposeAlleles takes its larger LDS size class
    (decompose_kernels.h DecompDims<4096>), multi-pass strips and full-matrix tracebacks carry the alignments; maxindel > 4096: the
    state lives in global memory (decompose_kernel_global) -- the reference has no limit (indigo.h:74, decompose.h:179-376)"""
    from indigo_oracle import decompose_trace
    from tracy_amd import capi, hostlib
    sigs, poss, refs, bcs = [], [], [], []
    for i, (seed, kind) in enumerate([(301, 0), (302, 0), (303, 2)]):
        ref, sig, pos, indel = hostlib.synth_decompose(seed, 5200, 2300, 40, kind | (16 if i == 1 else 0), 0.6)
        pri, sec, con, bcpos = hostlib.basecall(sig, pos, 0.33)
        assert len(pri) >= 2048
        sigs.append(sig); poss.append(pos); refs.append(ref); bcs.append((pri, sec, bcpos))
    profs = [hostlib.create_profile(sigs[i], bcs[i][2], bcs[i][0], bcs[i][1], 0, 0) for i in range(len(sigs))]
    for maxindel in (1000, 3000, 5000):
        hbc = capi.HostBaseCalls(sigs, [b[2] for b in bcs], [b[0] for b in bcs], [b[1] for b in bcs])
        got = ctx.decompose_traces(profs, hbc, refs, SC, maxindel=maxindel)
        for i in range(len(sigs)):
            w = decompose_trace(sigs[i], bcs[i][2], bcs[i][0], bcs[i][1], refs[i], SC, maxindel=maxindel)
            assert int(got["status"][i]) == w["status"] == 0
            assert got["primary"][i] == w["primary"] and got["secondary"][i] == w["secondary"], (maxindel, i)
            assert got["secdecomp_list"][i] == w["secdecomp"]
            assert got["dcp"][i] == w["dcp"], (maxindel, i)
            assert (float(got["fractions"][2 * i]), float(got["fractions"][2 * i + 1])) == w["af"]
            for k in range(3):
                assert int(got["score%d" % k][i]) == w["score%d" % k] and got["btr%d" % k][i] == w["btr%d" % k], (maxindel, i, k)


def test_staging_in_global_memory():
    """findBreakpoint / allelicFraction stage per-trace arrays in LDS; traces too long for it use a global scratch slice.  Forced
    here by lowering the limit (TRACYHIP_LDS_STAGE_LIMIT, read once per process): the chain tests in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRACYHIP_LDS_STAGE_LIMIT="1024")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_decompose.py"), "-m", "gpu", "-q", "-x",
                        "-k", "find_breakpoint or decompose_chain or decompose_traces_pipeline"], capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "3 passed" in r.stdout


def test_pruned_sweeps_of_decompose_where_they_certify_and_where_they_cannot(ctx, monkeypatch, capfd):
    """`tracy decompose` takes gotohScore(trace, window) of the voted strand and gotoh(allele, window) from the pruned sweep
    (front.h: prefix rows over the window, a certified band below them).  Windows that hold the locus twice cannot certify and are
    swept in full: the results are those of the run without the pruned sweep either way, and the oracle's (indigo.h:190-388)."""
    import re
    from indigo_oracle import decompose_trace
    from tracy_amd import capi, hostlib
    nd = 72
    d = hostlib.synth_decompose_batch(919, nd, 2200, 640, 0, mix=1)
    rng = np.random.default_rng(5)
    refs = []
    for i in range(nd):
        r = d["refs"][i].tobytes()
        if i % 3 == 0:  # the window twice, the second copy with a few substitutions: which copy wins is decided far below the prefix rows
            c = bytearray(r)
            for j in rng.integers(0, len(c), 12):
                c[int(j)] = int(rng.choice(list(b"ACGT")))
            r = r + bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(40, 300))).tolist()) + bytes(c)
        refs.append(r)

    def run():
        hbc = capi.HostBaseCalls([d["signal"][i] for i in range(nd)], [d["bcpos"][i] for i in range(nd)],
                                 [d["primary"][i].tobytes() for i in range(nd)], [d["secondary"][i].tobytes() for i in range(nd)])
        return ctx.decompose_traces([d["profiles"][i] for i in range(nd)], hbc, refs, SC)
    ctx.set_option("no_stream", 1)  # the tiers of the host-planned pipeline (stream-ordered: test_gpu_stream.py)
    pruned = run()
    said = ctx.last_call_stats()
    assert said["pruned"] >= nd // 2 and 8 <= said["pruned_uncertified"] < said["pruned"], said
    assert all(said["allele_pruned"][k] >= nd // 2 and 8 <= said["allele_uncertified"][k] <= said["allele_pruned"][k] - 8 for k in (0, 1)), said
    ctx.set_option("no_front", 1)
    plain = run()
    ctx.set_option("no_front", 0)
    ctx.set_option("no_stream", 0)
    assert int((np.asarray(pruned["status"]) == 0).sum()) > nd // 2
    for k in pruned:
        a, b = pruned[k], plain[k]
        if k in ("dcp_indel", "dcp_err", "ops"):
            continue
        if k == "bp":
            assert [(x.indelshift, x.traceleft, x.breakpoint, x.best_diff) for x in a] == [(x.indelshift, x.traceleft, x.breakpoint, x.best_diff) for x in b]
        elif isinstance(a, np.ndarray):
            assert np.array_equal(a, b), k
        elif isinstance(a, (list, tuple, dict, int, float, str, bytes)):
            assert a == b, k
        else:
            assert bytes(a) == bytes(b), k
    for i in range(0, 18):  # and the oracle, on windows of both kinds
        w = decompose_trace(d["signal"][i], d["bcpos"][i], d["primary"][i].tobytes(), d["secondary"][i].tobytes(), refs[i], SC)
        assert int(pruned["status"][i]) == w["status"], i
        if w["status"] != 0:
            continue
        for k in ("score_fwd", "score_rev", "forward", "score_trim"):
            assert int(pruned[k][i]) == int(w[k]), (i, k)
        for k in range(3):
            assert int(pruned["score%d" % k][i]) == w["score%d" % k], (i, k)
            assert pruned["btr%d" % k][i] == w["btr%d" % k], (i, k)
        for k in range(2):
            for nm in ("slice_begin", "slice_len", "ref_pos"):
                assert int(pruned["%s%d" % (nm, k)][i]) == int(w["%s%d" % (nm, k)]), (i, nm, k)


@pytest.mark.parametrize("sc", [(1, -1, -2, -1), (5, -4, -10, -1), (4, -6, -20, -8)])
def test_decompose_traces_with_other_scorings(ctx, sc):
    """tracyhip_decompose_traces under scorings other than tracy's default (indigo.h takes any, indigo.h:74-77): cheap mismatches make
    the shift scans of heterozygous traces come up empty, and "traverse the whole alignment" (decompose.h:327-343) then has dozens of
    basecalls to rewrite; ge = -1 switches the second certificate of the pruned sweeps off; dear gaps narrow every band."""
    from indigo_oracle import decompose_trace
    from tracy_amd import capi, hostlib
    nd = 24
    d = hostlib.synth_decompose_batch(777, nd, 2000, 650, 0, mix=1)
    hbc = capi.HostBaseCalls([d["signal"][i] for i in range(nd)], [d["bcpos"][i] for i in range(nd)],
                             [d["primary"][i].tobytes() for i in range(nd)], [d["secondary"][i].tobytes() for i in range(nd)])
    got = ctx.decompose_traces([d["profiles"][i] for i in range(nd)], hbc, [d["refs"][i].tobytes() for i in range(nd)], sc)
    fr = np.asarray(got["fractions"]).reshape(-1, 2)
    accepted = 0
    for i in range(nd):
        w = decompose_trace(d["signal"][i], d["bcpos"][i], d["primary"][i].tobytes(), d["secondary"][i].tobytes(), d["refs"][i].tobytes(), sc)
        assert int(got["status"][i]) == w["status"], i
        if w["status"] != 0:
            continue
        accepted += 1
        for k in ("score_fwd", "score_rev", "forward", "score_trim"):
            assert int(got[k][i]) == int(w[k]), (i, k)
        assert got["primary"][i] == w["primary"] and got["secondary"][i] == w["secondary"], i
        assert got["secdecomp_list"][i] == w["secdecomp"] and got["dcp"][i] == w["dcp"], i
        assert (float(fr[i, 0]), float(fr[i, 1])) == w["af"], i
        for k in range(3):
            assert int(got["score%d" % k][i]) == w["score%d" % k], (i, k)
            assert got["btr%d" % k][i] == w["btr%d" % k], (i, k)
        for k in range(2):
            for nm in ("slice_begin", "slice_len", "ref_pos"):
                assert int(got["%s%d" % (nm, k)][i]) == int(w["%s%d" % (nm, k)]), (i, nm, k)
    assert accepted >= nd // 2


def _random_decomp_cases(seed, count, long_window=False):
    """random gapped alignments with IUPAC secondaries, exotic reference letters, breakpoints anywhere (also behind the trace), small and
    large trims (tests/test_emu_decomp.py's generator); long_window: 1-2 kb of gap columns either side, as `tracy decompose` aligns them"""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(count):
        nb = int(rng.integers(60, 420)) if not long_window else int(rng.integers(500, 1100))
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
        r0, r1 = bytearray(), bytearray()
        lead = int(rng.integers(0, 30)) if not long_window else int(rng.integers(0, 1800))
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
        trail = int(rng.integers(0, 600)) if not long_window else int(rng.integers(0, 3900 - len(r0))) if len(r0) < 3900 else 0
        r0 += b"-" * trail
        r1 += bytes(rng.choice(list(b"ACGT"), size=trail).tolist())
        if it % 10 == 9:
            r1[len(r1) // 2] = ord("X")
        out.append(dict(rows=(bytes(r0), bytes(r1)), pri=bytes(pri), sec=bytes(sec), bp=orc.Breakpoint(1, 1, bpv, 0.5),
                        reflen=len(bytes(r1).replace(b"-", b"")), tl=tl, tr=tr))
    return out


@pytest.mark.parametrize("long_window", [False, True])
def test_decompose_wave_body_vs_oracle_and_step_kernel(ctx, long_window):
    """decomposeAlleles through the C ABI: the one-wave kernel with its working set in LDS (decompose_wave.h) -- incl. the traces it
    leaves to decompose_kernel (a breakpoint behind the trace, trims of every size) -- against the oracle and against option no_decomp_wave"""
    from tracy_amd import capi
    cases = _random_decomp_cases(99 if long_window else 98, 48, long_window)
    by_trim = {}
    for c in cases:
        by_trim.setdefault((c["tl"], c["tr"]), []).append(c)
    kinds = {}
    for (tl, tr), cs in by_trim.items():
        n = len(cs)
        sig = [np.zeros((4, 8), np.int32) for _ in cs]
        pos = [np.zeros(len(c["pri"]), np.int32) for c in cs]
        res = {}
        for mode in (0, 1):
            ctx.set_option("no_decomp_wave", mode)
            hbc = capi.HostBaseCalls(sig, pos, [c["pri"] for c in cs], [c["sec"] for c in cs])
            bps = [capi.Breakpoint(1, 1, c["bp"].breakpoint, 0.5) for c in cs]
            res[mode] = ctx.decompose_alleles(hbc, [c["rows"] for c in cs], bps, [c["reflen"] for c in cs], tl, tr, 1000, 5)
        ctx.set_option("no_decomp_wave", 0)
        assert res[0] == res[1]
        pri, sec, dcp, status = res[0]
        for i, c in enumerate(cs):
            w = orc.decompose_alleles(c["rows"][0], c["rows"][1], c["pri"], c["sec"], c["bp"], c["reflen"], tl, tr, 1000, 5)
            assert (pri[i], sec[i], dcp[i]) == (w[0], w[1], w[2]) and status[i][0] == w[3][0], (tl, tr, i)
            if w[3][0] == 1:
                assert tuple(status[i][:4]) == tuple(w[3][:4])
            kinds[w[3][0]] = kinds.get(w[3][0], 0) + 1
    assert len(kinds) == 3, kinds


def test_allelic_fraction_two_launch_form_vs_oracle_and_one_launch_kernel(ctx):
    """allelicFraction (decompose.h:412-621): af_prepare_kernel + af_search_kernel (the k of every (i, j) pair from the vertex of its
    parabola, exact sums through scalar loads) against the oracle's brute force and against the one-launch kernel, on realistic traces
    and on degenerate ones (no position with two plain bases, a single het position, all-zero signal, no het position at all)"""
    from tracy_amd import capi, hostlib
    rng = np.random.default_rng(4242)
    sigs, poss, pris, secs = [], [], [], []
    for seed, kind, frac in [(31, 0, 0.6), (32, 0, 0.35), (33, 1, 0.5), (34, 0, 0.8), (35, 0, 0.5), (36, 0, 0.2)]:
        ref, sig, pos, indel = hostlib.synth_decompose(seed, 1400, 480, 25, kind, frac)
        pri, sec, con, bcpos = hostlib.basecall(sig, pos, 0.33)
        sigs.append(sig); poss.append(bcpos); pris.append(pri); secs.append(sec)
    base_sig, base_pos, base_pri = sigs[0], poss[0], pris[0]
    n = len(base_pri)

    def variant(sec_fn, sig=None):
        sec = bytearray(base_pri)
        for i in range(n):
            sec[i] = sec_fn(i, base_pri[i])
        sigs.append(base_sig if sig is None else sig); poss.append(base_pos); pris.append(base_pri); secs.append(bytes(sec))

    other = {ord("A"): ord("C"), ord("C"): ord("G"), ord("G"): ord("T"), ord("T"): ord("A")}
    variant(lambda i, p: p)                                                      # no het position: (0.5, 0.5)
    variant(lambda i, p: ord("N") if 100 < i < 300 else p)                       # het positions, none with two plain bases: flat in k
    variant(lambda i, p: other.get(p, p) if i == 200 else p)                     # a single het position
    variant(lambda i, p: other.get(p, p) if i % 7 == 0 else p)                   # scattered SNV-like positions
    variant(lambda i, p: other.get(p, p) if 60 < i < 400 else p, np.zeros_like(base_sig))   # all-zero signal: 0 / 0 everywhere
    variant(lambda i, p: (ord("R") if i % 2 else other.get(p, p)) if 150 < i < 350 else p)  # IUPAC secondaries among plain ones
    for _ in range(6):                                                           # random signals: fractions anywhere in the grid
        sg = rng.integers(0, 2000, size=base_sig.shape).astype(np.int32)
        variant(lambda i, p: other.get(p, p) if 80 < i < 80 + int(rng.integers(5, 300)) else p, sg)
    hbc = capi.HostBaseCalls(sigs, poss, pris, secs)
    sd = np.frombuffer(b"".join(secs), dtype=np.uint8)
    res = {}
    for mode in (0, 1):
        ctx.set_option("no_af_split", mode)
        res[mode] = ctx.allelic_fraction(hbc, sd, 50, 50).copy()
    ctx.set_option("no_af_split", 0)
    assert np.array_equal(res[0], res[1], equal_nan=True), (res[0], res[1])
    seen = set()
    for i in range(len(sigs)):
        want = orc.allelic_fraction(sigs[i], poss[i], pris[i], secs[i], 50, 50)
        assert (float(res[0][i, 0]), float(res[0][i, 1])) == want, (i, res[0][i], want)
        seen.add(want)
    assert len(seen) >= 6 and (0.5, 0.5) in seen


def test_peak_table_instead_of_the_chromatogram(ctx):
    """tracyhip_basecalls::peaks (the four channels at every basecall's peak position, 16 bytes per basecall) instead of signal + bcpos:
    generateSecondaryDecomposed, allelicFraction (both launch forms) and the whole pipeline (stream-ordered and host-planned, one and
    two lanes) return what they return from the chromatograms -- which other tests hold against the oracle -- and the oracle's values"""
    from indigo_oracle import decompose_trace
    from sage_oracle import revcomp
    from tracy_amd import capi, hostlib
    sigs, poss, refs, bcs = [], [], [], []
    for i, (seed, kind, frac) in enumerate([(121, 0, 0.6), (122, 0, 0.45), (123, 1, 0.6), (124, 0, 0.7), (125, 1, 0.5), (126, 0, 0.3)]):
        ref, sig, pos, indel = hostlib.synth_decompose(seed, 1500, 520, 25, kind, frac)
        if i % 2:
            ref = revcomp(ref)
        pri, sec, con, bcpos = hostlib.basecall(sig, pos, 0.33)
        # IUPAC secondaries are what makes generateSecondaryDecomposed read the table (decompose.h:390-404): every ninth het position gets one
        sec = bytearray(sec)
        het = [q for q in range(len(sec)) if sec[q] != pri[q]]
        for z, q in enumerate(het[::9]):
            sec[q] = b"RYSWKM"[z % 6]
        sec = bytes(sec)
        sigs.append(sig); poss.append(pos); refs.append(ref); bcs.append((pri, sec, bcpos))
    assert any(any(ch in b"RYSWKM" for ch in b[1]) for b in bcs)
    profs = [hostlib.create_profile(sigs[i], bcs[i][2], bcs[i][0], bcs[i][1], 0, 0) for i in range(len(sigs))]

    def fresh():
        return capi.HostBaseCalls(sigs, [b[2] for b in bcs], [b[0] for b in bcs], [b[1] for b in bcs])
    h = fresh()
    tab = h.peak_table()
    assert tab.shape == (sum(len(b[0]) for b in bcs), 4) and int(tab[3, 2]) == int(sigs[0][2, bcs[0][2][3]])
    sd = {po: ctx.secondary_decomposed(fresh(), peaks_only=po).copy() for po in (False, True)}
    assert np.array_equal(sd[0], sd[1])
    for mode in (0, 1):
        ctx.set_option("no_af_split", mode)
        fr = {po: ctx.allelic_fraction(fresh(), sd[0], 50, 50, peaks_only=po).copy() for po in (False, True)}
        assert np.array_equal(fr[0], fr[1], equal_nan=True), mode
    ctx.set_option("no_af_split", 0)
    want = [decompose_trace(sigs[i], bcs[i][2], bcs[i][0], bcs[i][1], refs[i], SC) for i in range(len(sigs))]
    keys = ("score_fwd", "score_rev", "forward", "score_trim", "score0", "score1", "score2")
    for no_stream in (0, 1):
        for lanes in (1, 2):
            ctx.set_option("no_stream", no_stream)
            ctx.set_lanes(lanes)
            try:
                got = {po: ctx.decompose_traces(profs, fresh(), refs, SC, peaks_only=po) for po in (False, True)}
            finally:
                ctx.set_option("no_stream", 0)
                ctx.set_lanes(1)
            assert np.array_equal(got[0]["status"], got[1]["status"]), (no_stream, lanes)
            okt = [i for i in range(len(sigs)) if int(got[0]["status"][i]) == 0]  # (the later outputs of a trace that failed are unspecified)
            assert len(okt) >= 4
            for k in keys:
                for i in okt:
                    assert int(got[0][k][i]) == int(got[1][k][i]), (no_stream, lanes, k, i)
            for i in okt:
                for k in ("btr0", "btr1", "btr2", "primary", "secondary", "secdecomp_list", "dcp"):
                    assert got[0][k][i] == got[1][k][i], (no_stream, lanes, k, i)
                assert got[0]["fractions"][2 * i:2 * i + 2].tobytes() == got[1]["fractions"][2 * i:2 * i + 2].tobytes(), (no_stream, lanes, i)
            for i, w in enumerate(want):
                g = got[1]
                assert int(g["status"][i]) == w["status"]
                if w["status"] != 0:
                    continue
                assert g["primary"][i] == w["primary"] and g["secdecomp_list"][i] == w["secdecomp"] and g["dcp"][i] == w["dcp"], (no_stream, lanes, i)
                assert (float(g["fractions"][2 * i]), float(g["fractions"][2 * i + 1])) == w["af"], (no_stream, lanes, i)
                assert [g["btr%d" % k][i] for k in range(3)] == [w["btr%d" % k] for k in range(3)], (no_stream, lanes, i)
