"""Host stages of `tracy align` around the device pipeline: FASTA loading, trimTrace,
plotAlignment, alignmentTracePadding + JSON, the .align.fa writer -- the C++ in tracy_amd/host/sage_out.hpp
against the independent Python restatement in tests/sage_oracle.py, on alignments produced by the oracle."""
import os

import numpy as np
import pytest

import pyoracle as orc
import sage_oracle as so
from test_host_and_abi import make_trace

SC = (3, -5, -10, -4)


def consensus_ref(tr, pos, rng, flank=40, edits=True):
    """a reference carrying the trace's own primary calls (forward strand) with a few edits"""
    from tracy_amd import hostlib
    pri = hostlib.basecall(tr, pos, 0.33)[0]
    core = bytearray(pri.replace(b"N", b"A"))
    if edits and len(core) > 30:
        del core[10:13]                      # deletion in the reference -> gaps in row 1
        core[20:20] = b"ACGTAC"              # insertion in the reference -> gaps in row 0 (padding of the trace)
        core[25] = ord("T") if core[25] != ord("T") else ord("G")
    fl = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).tolist())
    return fl(flank) + bytes(core) + fl(flank)


@pytest.mark.parametrize("forward", [True, False])
def test_align_output_files(tmp_path, forward):
    from tracy_amd import hostlib
    rng = np.random.default_rng(11 + forward)
    for nb, linelimit in [(70, 60), (150, 60), (40, 25), (400, 60)]:
        tr, pos = make_trace(rng, nb, het=0.2)
        ref = consensus_ref(tr, pos, rng)
        pri, sec, con, bcpos, q = hostlib.basecall_qual(tr, pos, 0.33)
        full = hostlib.create_profile(tr, bcpos, pri, sec, 0, 0)
        refp = orc.create_profile_str(ref)
        score, btr = orc.gotoh_prof(full, refp, 1, 0, SC)
        row0, row1 = orc.create_alignment_prof(btr, full, refp)
        prefix = str(tmp_path / ("o%d" % nb))
        rpos = int(rng.integers(0, 5000))
        assert hostlib.align_outputs(prefix, "mytrace", tr, pos, 0.33, row0, row1, "chrT", ref, rpos, forward, score, linelimit) == 0
        want_txt = so.plot_alignment(row0, row1, "chrT", rpos, len(ref), forward, score, linelimit)
        assert open(prefix + ".txt").read() == want_txt
        padded = so.alignment_trace_padding(row0, tr, bcpos, pri, sec, con, q)
        assert open(prefix + ".json").read() == so.trace_align_json(padded, "chrT", rpos, forward, row0, row1)
        assert open(prefix + ".align.fa").read() == so.align_fasta_text("mytrace", "chrT", forward, row0, row1)
        assert row0.count(b"-") > 0 and b"-99" in open(prefix + ".json", "rb").read()


def test_plot_alignment_known_answer(tmp_path):
    """hand-checked small case (reverse strand numbering counts down from pos + len)"""
    txt = so.plot_alignment(b"AC-GT", b"ACTG-", "c", 10, 4, False, 7, 60)
    lines = txt.split("\n")
    assert lines[0] == ">Alt" and lines[1] == "ACGT" and lines[2] == ">Ref c:11-14 reversecomplement" and lines[3] == "ACTG"
    assert "Alt         1 AC-GT" in lines and "Ref        14 ACTG-" in lines and " " * 14 + "|| | " in lines


def test_trim_trace_matches_restatement():
    from tracy_amd import hostlib
    rng = np.random.default_rng(3)
    hits = 0
    for it in range(14):
        nb = int(rng.integers(30, 500))
        tr, pos = make_trace(rng, nb, het=0.05)
        # noisy ends: mixed peaks make ambiguous secondary calls
        k = int(rng.integers(5, nb // 3))
        tr2, _ = make_trace(rng, nb, het=0.9)
        tr[:, :pos[k]] = tr2[:, :pos[k]]
        tr[:, pos[nb - k]:] = tr2[:, pos[nb - k]:]
        pri, sec, con, bcpos = hostlib.basecall(tr, pos, 0.33)
        for stringency in (1, 2, 5, 9):
            got = hostlib.trim_trace(tr, pos, 0.33, stringency)
            want = so.trim_trace(stringency, sec, bcpos.tolist())
            assert got == want, (it, stringency)
            hits += got != (0, 0)
    assert hits > 10


def test_load_single_fasta(tmp_path):
    from tracy_amd import hostlib
    cases = {
        "a.fa": b">chr1 some (odd) name: x#1\nacgtn\nRYKMSWBDHV\n\nACGT\n",
        "b.fa": b">w\r\nACGT\r\nTT\r\n",
        "c.fa": b">one\nACGT\n>two\nAC\n",
        "d.fa": b">gap\nAC-GT\n",
        "e.fa": b"ACGT\n>late\nTT\n",
    }
    for name, data in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        assert hostlib.load_fasta(p) == so.load_single_fasta(p), name
    assert hostlib.load_fasta(str(tmp_path / "a.fa")) == ("chr1 some odd name x1", "ACGTNNNNNNNNNNNACGT")
    assert hostlib.load_fasta(str(tmp_path / "c.fa")) is None and hostlib.load_fasta(str(tmp_path / "d.fa")) is None
