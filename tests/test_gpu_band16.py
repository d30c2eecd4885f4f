"""tracyhip_gotoh_banded (band kernels, tracy_amd/csrc/band16.h) on the GPU against the oracle: strings and profile rows, free
and paid end gaps, both strands of the origin-tracking sweep, bands chosen as the pipelines choose them."""
import random

import numpy as np
import pytest

import pyoracle as orc

pytestmark = pytest.mark.gpu
SC = (3, -5, -10, -4)
COMP = bytes.maketrans(b"ACGT", b"TGCA")


@pytest.fixture(scope="module")
def ctx():
    import tracy_amd
    c = tracy_amd.Context(0)
    yield c
    c.close()


def rand_seq(rng, n):
    return bytes(rng.choice(b"ACGT") for _ in range(n))


def mutate(s, rate, rng):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice(b"ACGT")); out.append(ch)
        elif x < rate:
            out.append(rng.choice(b"ACGTN"))
        else:
            out.append(ch)
    return bytes(out) or b"A"


def ends_of(btr, n):
    fwd = btr[::-1]
    return len(fwd) - len(fwd.lstrip(b"h")), n - (len(fwd) - len(fwd.rstrip(b"h")))


@pytest.mark.parametrize("hfree", [1, 0])
def test_banded_strings_equal_the_whole_matrix_when_the_band_holds_the_path(ctx, hfree):
    rng = random.Random(100 + hfree)
    a1, a2, lo, hi, wants = [], [], [], [], []
    for it in range(600):
        m = rng.randint(1, 1100 if it % 7 == 0 else 300)
        a = rand_seq(rng, m)
        core = mutate(a, rng.choice([0.0, 0.02, 0.08]), rng)
        b = (rand_seq(rng, rng.randint(0, 60)) + core + rand_seq(rng, rng.randint(0, 60))) if hfree else core
        ws, wb = orc.gotoh_str(a, b, hfree, 0, SC)
        n = len(b)
        if hfree:  # the band the decompose pipeline uses: around the end of the alignment, as wide as the score allows gap steps
            lead, ce = ends_of(wb, n)
            g = (SC[0] * m - ws) // 4
            d1 = ce - m
            dl, dh = d1 - g - 1, d1 + g + 1
        else:
            g = (SC[0] * m - ws) // 11 + 2
            dl, dh = -g - max(0, m - n), g + max(0, n - m)
        if dh - dl > 170:
            continue
        a1.append(a); a2.append(b); lo.append(dl); hi.append(dh); wants.append((ws, wb))
    prm = SC + (hfree, 0)
    sc, btr = ctx.align_banded(a1, a2, prm, lo, hi)
    assert len(wants) > 400
    for i, (ws, wb) in enumerate(wants):
        assert (int(sc[i]), btr[i]) == (ws, wb), (i, len(a1[i]), len(a2[i]), lo[i], hi[i])


def test_banded_origin_sweep(ctx):
    rng = random.Random(7)
    a1, a2, lo, hi, wants = [], [], [], [], []
    for it in range(500):
        m = rng.randint(5, 1000 if it % 5 == 0 else 250)
        a = rand_seq(rng, m)
        b = rand_seq(rng, rng.randint(0, 300)) + mutate(a, rng.choice([0.0, 0.05]), rng) + rand_seq(rng, rng.randint(0, 300))
        ws, wb = orc.gotoh_str(a, b, 1, 0, SC)
        lead, ce = ends_of(wb, len(b))
        g = (SC[0] * m - ws) // 4
        d1 = ce - m
        if ce == 0 or 2 * g + 2 > 170:
            continue
        a1.append(a); a2.append(b); lo.append(d1 - g - 1); hi.append(d1 + g + 1); wants.append((ws, (lead, ce)))
    sc, ends = ctx.align_banded(a1, a2, SC + (1, 0), lo, hi, origin=True)
    assert len(wants) > 300
    for i, (ws, we) in enumerate(wants):
        assert (int(sc[i]), ends[i]) == (ws, we), (i, len(a1[i]), len(a2[i]))


def test_banded_profile_rows(ctx):
    """gotoh(trace profile, _createProfile(slice)) on a band, as the final alignments of `tracy align` (sage.h:260)"""
    from tracy_amd import hostlib
    refs, profs, rev = hostlib.synth_align(4321, 40, 1400, 1000, 0)
    a1, a2, lo, hi, wants = [], [], [], [], []
    for i in range(40):
        ref = refs[i].tobytes()
        view = ref[::-1].translate(COMP) if rev[i] else ref
        p = np.ascontiguousarray(profs[i][:, :600 + 10 * i])
        m = p.shape[1]
        ws, wb = orc.gotoh_prof(p, orc.create_profile_str(view), 1, 0, SC)
        lead, ce = ends_of(wb, len(view))
        sl = view[max(0, lead - 50):min(len(view), ce + 50)]
        ws, wb = orc.gotoh_prof(p, orc.create_profile_str(sl), 1, 0, SC)
        W = 35  # (the synthetic traces lose < 4 * 35 against their row maxima)
        a1.append(p); a2.append(sl); lo.append(-W - max(0, m - len(sl))); hi.append(W + max(0, len(sl) - m)); wants.append((ws, wb))
    sc, btr = ctx.align_banded(a1, a2, SC + (1, 0), lo, hi)
    for i, (ws, wb) in enumerate(wants):
        assert (int(sc[i]), btr[i]) == (ws, wb), i


def test_banded_string_rows_outside_the_five_letters_are_refused(ctx):
    """the string tables of the band kernels know A C G T N: a row with any other byte (lower case, IUPAC, '-') is refused with
    TRACYHIP_ERR_ARG instead of being scored as a mismatch against an identical column byte (gotoh.h compares bytes, align.h:96-101);
    columns may hold anything, and the same rows in upper case go through"""
    from tracy_amd import capi
    good = b"ACGTNACGTTGCA" * 8
    col = b"ACGTRYacgt-NACGTTGCA" * 6  # columns: any byte (they mismatch every row letter they are not)
    lo, hi = [-(len(good))], [len(col)]
    band = (max(lo[0], -60), min(hi[0], 60))
    sc, btr = ctx.align_banded([good], [col], SC + (1, 0), [band[0]], [band[1]])
    assert len(btr[0]) > 0 or int(sc[0]) <= 0
    for bad in (good[:5] + b"a" + good[6:], good[:9] + b"R" + good[10:], good[:3] + b"-" + good[4:]):
        with pytest.raises(capi.TracyHipError) as e:
            ctx.align_banded([bad], [col], SC + (1, 0), [band[0]], [band[1]])
        assert e.value.code == capi.ERR_ARG and "A C G T N" in str(e.value)
