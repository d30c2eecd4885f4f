"""The pruned orientation sweep of `tracy align` (tracy_amd/csrc/front.h: prefix rows over the whole window, a certified band below
them) against the oracle's sage.h chain on the traces it is built for AND on the ones whose certificate must fail -- the target
twice in the window, a chimeric trace, a long insertion below the prefix rows, windows that end inside the alignment: every field
and the alignment string are the reference's either way, with both exact scores and with the strand decided by certificate."""
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SC = (3, -5, -10, -4)
COMP = bytes.maketrans(b"ACGT", b"TGCA")
FIELDS = ("forward", "score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final")


@pytest.fixture(scope="module")
def ctx():
    import tracy_amd
    c = tracy_amd.Context(0)
    yield c
    c.close()


def rand_seq(rng, n):
    return bytes(rng.choice(list(b"ACGT"), size=n).tolist())


def noisy(rng, s, rate):
    out = bytearray()
    for ch in s:
        u = rng.random()
        if u < rate / 3:
            continue
        if u < 2 * rate / 3:
            out.append(int(rng.choice(list(b"ACGT"))))
        out.append(int(rng.choice(list(b"ACGT"))) if u > 1 - rate / 3 else ch)
    return bytes(out)


def profile_of(rng, seq, sharp=0.85):
    idx = {65: 0, 67: 1, 71: 2, 84: 3}
    p = np.zeros((6, len(seq)), np.float32)
    for j, ch in enumerate(seq):
        main = np.float32(rng.uniform(sharp, 1.0))
        rest = rng.random(3).astype(np.float32)
        rest = rest / rest.sum() * (np.float32(1) - main)
        col = np.zeros(4, np.float32)
        col[idx[ch]] = main
        col[[k for k in range(4) if k != idx[ch]]] = rest
        p[:4, j] = col
    return p


def cases(rng):
    """(profile, window, kind)"""
    out = []
    for it in range(64):
        mf = int(rng.choice([330, 520, 700, 880, 1010]))  # strip heights 8 / 12 / 15 / 16 in one batch
        seq = rand_seq(rng, mf)
        kind = ["plain", "plain", "twice", "chimera", "insertion", "deletion", "cut", "edge"][it % 8]
        fl = lambda k: rand_seq(rng, int(k))  # noqa: E731
        if kind == "plain":
            win = fl(rng.integers(0, 1500)) + noisy(rng, seq, rng.choice([0.0, 0.02, 0.06])) + fl(rng.integers(0, 1500))
        elif kind == "twice":  # two copies: which one is better is decided below the prefix rows
            head = noisy(rng, seq[:200], 0.01)
            a = head + noisy(rng, seq[200:], 0.10)
            b = head + noisy(rng, seq[200:], 0.01)
            if rng.random() < 0.5:
                a, b = b, a
            win = fl(rng.integers(0, 300)) + a + fl(rng.integers(50, 400)) + b + fl(rng.integers(0, 300))
        elif kind == "chimera":  # the trace follows the window for 300 bases only
            win = fl(rng.integers(0, 900)) + noisy(rng, seq[:300], 0.02) + fl(rng.integers(600, 1500))
        elif kind == "insertion":
            cut = int(rng.integers(180, mf - 100))
            win = fl(rng.integers(0, 900)) + seq[:cut] + fl(rng.integers(100, 260)) + seq[cut:] + fl(rng.integers(0, 900))
        elif kind == "deletion":
            cut = int(rng.integers(180, mf - 150))
            win = fl(rng.integers(0, 900)) + seq[:cut] + seq[cut + int(rng.integers(100, 140)):] + fl(rng.integers(0, 900))
        elif kind == "cut":  # the window ends inside the alignment
            win = fl(rng.integers(0, 900)) + seq[:int(rng.integers(200, mf - 20))]
        else:  # the alignment starts at the first column of the window / the window is hardly longer than the trace
            win = noisy(rng, seq, 0.02) + fl(rng.integers(0, 30))
        if rng.random() < 0.5:  # a reverse trace: the window holds the reverse complement
            win = win.translate(COMP)[::-1]
        if rng.random() < 0.2:
            w = bytearray(win)
            w[int(rng.integers(0, len(w)))] = ord("N")
            win = bytes(w)
        out.append((profile_of(rng, seq), win, kind))
    return out


def test_pruned_sweep_equals_the_reference_where_it_certifies_and_where_it_cannot(ctx, monkeypatch, capfd):
    import sage_oracle as so
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(77)
    cs = cases(rng)
    profs, wins = [c[0] for c in cs], [c[1] for c in cs]
    with ThreadPoolExecutor(16) as pool:
        want = list(pool.map(lambda i: so.align_trace(profs[i], wins[i], SC, 50, 50), range(len(cs))))
    ctx.set_option("no_stream", 1)  # the tiers of the host-planned pipeline (the stream-ordered form of this batch: test_gpu_stream.py)
    exact = ctx.align_traces(profs, wins, SC, 50, 50)
    said = ctx.last_call_stats()
    fast = ctx.align_traces(profs, wins, SC, 50, 50, exact_scores=False)
    pruned, notcert = said["pruned"], said["pruned_uncertified"]
    assert pruned >= 48 and 8 <= notcert <= pruned - 16, said  # both outcomes are exercised
    for i, w in enumerate(want):
        for k in FIELDS + ("score_fwd", "score_rev"):
            assert int(exact[k][i]) == int(w[k]), (i, cs[i][2], k)
        assert exact["btr"][i] == w["btr"], (i, cs[i][2])
        for k in FIELDS:
            assert int(fast[k][i]) == int(w[k]), (i, cs[i][2], k)
        assert fast["btr"][i] == w["btr"], (i, cs[i][2])
        win, lose = ("score_fwd", "score_rev") if int(w["forward"]) else ("score_rev", "score_fwd")
        assert int(fast[win][i]) == int(w[win]) and int(fast[lose][i]) >= int(w[lose]), (i, cs[i][2])
    # the same batch without the pruned sweep: nothing but the work differs
    ctx.set_option("no_front", 1)
    plain = ctx.align_traces(profs, wins, SC, 50, 50)
    ctx.set_option("no_front", 0)
    ctx.set_option("no_stream", 0)
    for k in FIELDS + ("score_fwd", "score_rev"):
        assert plain[k].tolist() == exact[k].tolist(), k
    assert plain["btr"] == exact["btr"]


def test_heterozygous_traces_certify_by_the_second_bound(ctx, monkeypatch, capfd):
    """A trace downstream of a heterozygous indel shows two equal peaks per position: such a row scores -1 against either base
    while the first certificate allows it 0, which over ~700 rows is more than any band pays for.  front_certify_body's second
    bound (rows allowed max(row maximum, -1)) certifies them; results are the oracle's."""
    import sage_oracle as so
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(41)
    idx = {65: 0, 67: 1, 71: 2, 84: 3}
    profs, wins = [], []
    for it in range(24):
        mf = int(rng.integers(850, 1010))
        seq = rand_seq(rng, mf)
        bp = int(rng.integers(150, 400))
        shift = int(rng.integers(1, 12))
        alt = seq[:bp] + seq[bp + shift:] + rand_seq(rng, shift)
        p = np.zeros((6, mf), np.float32)
        for j in range(mf):
            a, b = idx[seq[j]], idx[alt[j]]
            if a == b:
                p[a, j] = 1.0
            else:
                p[a, j] = p[b, j] = 0.5
        win = rand_seq(rng, int(rng.integers(0, 1500))) + noisy(rng, seq, 0.01) + rand_seq(rng, int(rng.integers(0, 1500)))
        if it % 2:
            win = win.translate(COMP)[::-1]
        profs.append(p); wins.append(win)
    with ThreadPoolExecutor(16) as pool:
        want = list(pool.map(lambda i: so.align_trace(profs[i], wins[i], SC, 50, 50), range(len(profs))))
    for no_stream in (1, 0):  # planned by the host, planned on the device: the same certificates
        ctx.set_option("no_stream", no_stream)
        got = ctx.align_traces(profs, wins, SC, 50, 50)
        said = ctx.last_call_stats()
        assert said["stream_ordered"] == 1 - no_stream and said["pruned"] >= 20 and said["pruned_uncertified"] <= 4, said
        for i, w in enumerate(want):
            for k in FIELDS + ("score_fwd", "score_rev"):
                assert int(got[k][i]) == int(w[k]), (i, k, no_stream)
            assert got["btr"][i] == w["btr"], (i, no_stream)
