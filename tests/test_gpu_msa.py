"""BASELINE configs[4] at test size: all-pairs profile x profile scores, UPGMA, progressive alignment, consensus and
strand assignment of `tracy assemble` (tracy_amd/host/msa.hpp, DPs on the GPU) against the Python restatement over
the oracle (tests/msa_oracle.py)."""
import numpy as np
import pytest

import msa_oracle as mo
import pyoracle as orc
import sage_oracle as so

pytestmark = pytest.mark.gpu
SC = (3, -5, -10, -4)


def overlapping_profiles(rng, n, region=900, tlen=260, rc_some=False):
    """noisy trace-like profiles of overlapping stretches of one region"""
    ref = bytes(rng.choice(list(b"ACGT"), size=region).tolist())
    profs = []
    for i in range(n):
        start = int(i * (region - tlen) / max(n - 1, 1))
        seq = bytearray(ref[start:start + tlen])
        for k in range(len(seq)):
            if rng.random() < 0.02:
                seq[k] = int(rng.choice(list(b"ACGT")))
        if rng.random() < 0.5:
            del seq[50:52]
        p = np.zeros((6, len(seq)), np.float32)
        for j, ch in enumerate(bytes(seq)):
            col = rng.random(4).astype(np.float32) * np.float32(0.08)
            col[b"ACGT".index(ch)] += np.float32(1.0)
            p[:4, j] = col / col.sum()
        if rc_some and i % 3 == 1:
            p = orc.revcomp_profile(p)
        profs.append(np.ascontiguousarray(p))
    return profs


@pytest.fixture(scope="module")
def ctx():
    import tracy_amd
    c = tracy_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n", [1, 2, 7])
def test_msa_matches_the_restatement(ctx, n):
    from tracy_amd import msalib
    rng = np.random.default_rng(50 + n)
    profs = overlapping_profiles(rng, n)
    rows, sidx = msalib.msa(ctx, profs, SC)
    want_rows, want_sidx = mo.msa(profs, SC)
    assert sidx == want_sidx
    assert [r.decode() for r in rows] == want_rows
    g, cs, q = msalib.consensus(rows, 0.5, False)
    wg, wcs, wq = mo.consensus(want_rows, 0.5, False)
    assert (g.decode(), cs.decode(), q.decode()) == (wg, wcs, wq)
    if n > 2:
        g2 = msalib.consensus(rows, 0.9, True)
        assert tuple(x.decode() for x in g2) == mo.consensus(want_rows, 0.9, True)
        prof = msalib.profile_of_alignment(rows)
        assert np.array_equal(prof.view(np.uint32), mo.profile_of_alignment(want_rows).view(np.uint32))


def test_unrelated_sequences_stay_unmerged(ctx):
    """negative scores are never picked by closestPair (msa.h:47-58): the result is the last sequence alone"""
    from tracy_amd import msalib
    rng = np.random.default_rng(9)
    profs = []
    for _ in range(3):
        seq = rng.integers(0, 4, size=120)
        p = np.zeros((6, 120), np.float32)
        p[seq, np.arange(120)] = 1
        profs.append(p)
    rows, sidx = msalib.msa(ctx, profs, (3, -5, -100, -40))
    want_rows, want_sidx = mo.msa(profs, (3, -5, -100, -40))
    assert sidx == want_sidx and [r.decode() for r in rows] == want_rows


def test_strand_assignment(ctx):
    from tracy_amd import msalib
    rng = np.random.default_rng(77)
    profs = overlapping_profiles(rng, 6, region=700, tlen=240, rc_some=True)
    got, fwd = msalib.rev_seq_based_on_dist(ctx, profs, SC)
    want, wfwd = mo.rev_seq_based_on_dist(profs, SC)
    assert fwd == wfwd and not all(fwd)
    for a, b in zip(got, want):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
