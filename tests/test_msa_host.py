"""Host-side pieces of the multiple alignment (tracy_amd/host/msa.hpp: _createProfile(char MSA), consensus) against the
Python restatement -- no GPU needed.  The device-backed msa() / revSeqBasedOnDist() are covered by tests/test_gpu_msa.py."""
import numpy as np

import msa_oracle as mo


def random_alignment(rng, nrow, ncol):
    rows = []
    for i in range(nrow):
        r = rng.choice(list("ACGTacgtNn-X"), size=ncol, p=[.2, .2, .2, .2, .01, .01, .01, .01, .01, .01, .11, .03]).tolist()
        lead, trail = int(rng.integers(0, ncol // 2)), int(rng.integers(0, ncol // 3))
        for j in range(lead):
            r[j] = "-"
        for j in range(ncol - trail, ncol):
            r[j] = "-"
        rows.append("".join(r))
    if nrow > 2:
        rows[1] = "-" * ncol  # a row of gaps only
    return rows


def test_profile_of_alignment_and_consensus():
    from tracy_amd import msalib
    rng = np.random.default_rng(8)
    for nrow, ncol in [(1, 30), (2, 80), (5, 200), (9, 333)]:
        rows = random_alignment(rng, nrow, ncol)
        braw = [r.encode() for r in rows]
        got = msalib.profile_of_alignment(braw)
        want = mo.profile_of_alignment(rows)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        for frac, ign in [(0.5, False), (0.0, False), (1.0, False), (0.34, True)]:
            if ign and nrow < 2:
                continue
            g = msalib.consensus(braw, frac, ign)
            assert tuple(x.decode() for x in g) == mo.consensus(rows, frac, ign), (nrow, ncol, frac, ign)


def test_upgma_truncating_average():
    """the averaged distance is a C++ integer division (toward zero), msa.h:64"""
    d = [[-1] * 7 for _ in range(7)]
    d[0][1], d[0][2], d[1][2] = 10, 3, -4
    root, p = mo.upgma(d, 3)
    # (3 + -4) / 2 == 0 in C++ (not -1 as Python's floor division): 0 > -1, so sequence 2 still joins the tree
    assert p[3][1:] == [0, 1] and p[4][1:] == [2, 3] and root == 4
