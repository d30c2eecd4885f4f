"""CPU-side checks: the C ABI library loads and exports every symbol include/tracy_hip.h declares; the
host C++ stages (basecall, createProfile) match the oracle and, for basecall, the reference's own abif.h
(oracle/_ref) and the committed golden vectors generated from it."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "tracy_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tracyhip_[a-z_0-9]+)\s*\(", txt)))


def test_abi_exports_every_declared_symbol():
    from tracy_amd import capi
    lib = capi.lib()
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.tracyhip_version()


def test_no_device_is_a_loud_error():
    """without a GPU every compute entry point must fail (no CPU fallback)"""
    import torch
    from tracy_amd import capi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.TracyHipError) as e:
        capi.Context(0)
    assert e.value.code == capi.ERR_NODEVICE


def make_trace(rng, nb, het=0.0):
    samples = 12 * nb + 12
    tr = np.zeros((4, samples), dtype=np.int32)
    pos = 6 + 12 * np.arange(nb, dtype=np.int32)
    x = np.arange(-5, 6)
    tri = 1.0 - np.abs(x) / 6.0
    for j in range(nb):
        amp = rng.uniform(300, 1200)
        b = int(rng.integers(0, 4))
        tr[b, pos[j] - 5:pos[j] + 6] += (amp * tri).astype(np.int32)
        if rng.random() < het:
            b2 = (b + int(rng.integers(1, 4))) % 4
            tr[b2, pos[j] - 5:pos[j] + 6] += (amp * rng.uniform(0.3, 1.0) * tri).astype(np.int32)
        if rng.random() < het / 3:
            b3 = int(rng.integers(0, 4))
            tr[b3, pos[j] - 5:pos[j] + 6] += (amp * rng.uniform(0.3, 0.8) * tri).astype(np.int32)
        bg = int(rng.integers(0, 4))
        tr[bg, pos[j] - 5:pos[j] + 6] += (amp * rng.uniform(0.02, 0.2) * tri).astype(np.int32)
    jitter = rng.integers(-2, 3, size=nb).astype(np.int32)
    return tr, (pos + jitter).astype(np.int32)


def test_host_basecall_and_profile_match_oracle_and_reference():
    from tracy_amd import hostlib
    rng = np.random.default_rng(17)
    ref = orc.ref_lib()
    for it in range(12):
        tr, pos = make_trace(rng, int(rng.integers(30, 400)), het=[0.0, 0.3, 0.8][it % 3])
        got = hostlib.basecall(tr, pos, 0.33)
        want = orc.basecall(tr, pos, 0.33)
        for a, b in zip(got, want):
            assert np.array_equal(np.frombuffer(a, np.uint8) if isinstance(a, bytes) else a,
                                  np.frombuffer(b, np.uint8) if isinstance(b, bytes) else b)
        if ref is not None:  # the reference's own abif.h, compiled in this container only
            rr = orc.ref_basecall(tr, pos, 0.33)
            assert rr[0] == got[0] and rr[1] == got[1] and rr[2] == got[2] and np.array_equal(rr[3], got[3])
        pri, sec, con, bcpos = got
        for (tl, trr) in [(0, 0), (5, 7), (1000, 1000)]:
            p_host = hostlib.create_profile(tr, bcpos, pri, sec, tl, trr)
            p_orc = orc.create_profile_trace(tr, bcpos, pri, sec, tl, trr)
            assert p_host.shape == p_orc.shape and np.array_equal(p_host.view(np.uint32), p_orc.view(np.uint32))


def test_basecall_golden_vectors_from_reference():
    """tests/golden/abif_basecall.json was produced by tests/golden/make_abif_golden.py from the
    reference's abif.h (oracle/_ref); both the oracle and the host library must reproduce it."""
    from tracy_amd import hostlib
    path = os.path.join(ROOT, "tests", "golden", "abif_basecall.json")
    cases = json.load(open(path))
    assert len(cases) >= 6
    for c in cases:
        tr = np.array(c["trace"], dtype=np.int32)
        pos = np.array(c["basecallpos"], dtype=np.int32)
        for fn in (orc.basecall, hostlib.basecall):
            pri, sec, con, bcpos = fn(tr, pos, c["sigratio"])
            assert pri.decode() == c["primary"] and sec.decode() == c["secondary"] and con.decode() == c["consensus"]
            assert bcpos.tolist() == c["bcPos"]
    for a, b, want in json.load(open(os.path.join(ROOT, "tests", "golden", "abif_iupac.json"))):
        assert orc.lib().orc_iupac2(a.encode(), b.encode()) == want.encode()
        assert hostlib.lib().tracyhost_iupac(a.encode(), b.encode()) == want.encode()


def test_synth_workload_is_seeded_and_plausible():
    from tracy_amd import hostlib
    r1, p1, v1 = hostlib.synth_align(1000, 6, 2000, 300, 2)
    r2, p2, v2 = hostlib.synth_align(1000, 6, 2000, 300, 3)
    assert np.array_equal(r1, r2) and np.array_equal(p1, p2) and np.array_equal(v1, v2)
    assert set(np.unique(r1).tolist()) <= set(b"ACGT")
    assert np.allclose(p1[:, :4].sum(axis=1), 1.0, atol=1e-5)
    assert (p1[:, :4].max(axis=1) > 0.8).mean() > 0.95


def test_breakpoint_selection_by_reductions_equals_the_sequential_walk():
    """findBreakpoint keeps a float-typed running maximum while it compares doubles (decompose.h:27-55).  The kernel answers
    with three reductions (largest float, first position that rounds to it, last position strictly above it): the same
    position as the sequential walk, also where many values share one float and differ as doubles"""
    import numpy as np
    rng = np.random.default_rng(31)

    def walk(diff):
        best, idx = np.float32(0), None
        for i, v in enumerate(diff):
            if v > float(best):
                idx, best = i, np.float32(v)
        return idx, best

    def reduced(diff):
        g = diff.astype(np.float32)
        pos = diff > 0
        F = np.float32(max(np.float32(0), g[pos].max())) if pos.any() else np.float32(0)
        above = np.nonzero(diff > float(F))[0]
        if len(above):
            return int(above[-1]), F
        first = np.nonzero(pos & (g == F))[0] if F > 0 else []
        return (int(first[0]) if len(first) else None), F

    for trial in range(3000):
        n = int(rng.integers(1, 60))
        kind = trial % 4
        if kind == 0:
            diff = rng.random(n)
        elif kind == 1:  # a few floats, each met by several doubles around it
            base = rng.choice(np.float32([0.25, 0.5, 0.75, 1.0, 0.1]).astype(np.float64), n)
            diff = base * (1 + (rng.integers(-3, 4, n) * 2.0 ** -30))
        elif kind == 2:
            diff = np.where(rng.random(n) < 0.3, 0.0, rng.choice([1e-46, -1.0, 0.5, 0.5 + 2.0 ** -40, 0.5 - 2.0 ** -40], n))
        else:
            diff = np.float64(np.float32(rng.random())) + rng.integers(-2, 3, n) * 2.0 ** -28
        diff = np.asarray(diff, np.float64)
        wi, wb = walk(diff)
        ri, rb = reduced(diff)
        assert wi == ri and wb == rb, (trial, diff.tolist())
