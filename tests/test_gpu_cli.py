"""BASELINE.json configs[0] end to end on the GPU box: a synthetic ~800-base ABIF written by the build's own
writer, `tracy_amd_cli align` against a 5 kb FASTA (and against a wildtype trace, and in --batch mode), all
four output files compared byte for byte with the oracle chain (tests/sage_oracle.py)."""
import os
import subprocess

import numpy as np
import pytest

import pyoracle as orc
import sage_oracle as so
from test_host_and_abi import make_trace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tracy_amd", "bin", "tracy_amd_cli")
SC = (3, -5, -10, -4)
pytestmark = pytest.mark.gpu


def synth_case(rng, tmp, tag, nb=800, nref=5000, reverse=False, wildtype=False):
    """trace file + reference file; the reference carries the trace's calls (with a few edits) at a random offset"""
    from tracy_amd import hostlib
    tr, pos = make_trace(rng, nb, het=0.03)
    tr = np.minimum(tr, 32000)
    pri = hostlib.basecall(tr, pos, 0.33)[0]
    trace_path = os.path.join(tmp, tag + ".ab1")
    hostlib.write_abif(trace_path, tr, pos, pri[:len(pos)].ljust(len(pos), b"N"), np.full(len(pos), 40, np.uint8))
    core = bytearray(pri.replace(b"N", b"C"))
    del core[300:304]
    core[500:500] = b"GATTACA"
    core[100] = ord("A") if core[100] != ord("A") else ord("C")
    if wildtype:
        # wildtype chromatogram: one clean peak per base of the edited sequence
        wt = np.zeros((4, 12 * len(core) + 12), np.int32)
        wpos = 6 + 12 * np.arange(len(core), dtype=np.int32)
        tri = (900 * (1.0 - np.abs(np.arange(-5, 6)) / 6.0)).astype(np.int32)
        seq = bytes(core) if not reverse else so.revcomp(bytes(core))
        for j, ch in enumerate(seq):
            wt[b"ACGT".index(ch), wpos[j] - 5:wpos[j] + 6] += tri
        ref_path = os.path.join(tmp, tag + "_wt.ab1")
        hostlib.write_abif(ref_path, wt, wpos, seq, np.full(len(seq), 40, np.uint8))
    else:
        off = int(rng.integers(200, nref - len(core) - 200))
        fl = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).tolist())
        ref = fl(off) + bytes(core) + fl(nref - off - len(core))
        if reverse:
            ref = so.revcomp(ref)
        ref_path = os.path.join(tmp, tag + ".fa")
        with open(ref_path, "w") as f:
            f.write(">chrSyn test\n")
            for i in range(0, len(ref), 70):
                f.write(ref[i:i + 70].decode().lower() + "\n")
    return trace_path, ref_path


def expected_files(trace_path, ref_path, stem, trim=(50, 50), linelimit=60):
    from tracy_amd import hostlib
    t = hostlib.read_trace(trace_path)
    tr, pos = t["signal"], t["basecallpos"]
    pri, sec, con, bcpos, q = hostlib.basecall_qual(tr, pos, 0.33)
    full = orc.create_profile_trace(tr, bcpos, pri, sec, 0, 0)
    if ref_path.endswith(".fa"):
        name, ref = so.load_single_fasta(ref_path)
        r = so.align_trace(full, ref.encode(), SC, trim[0], trim[1])
        refslice, refp = r["refslice"], orc.create_profile_str(r["refslice"])
        forward, rpos, score, btr = bool(r["forward"]), r["ref_pos"], r["score_final"], r["btr"]
    else:
        g = hostlib.read_trace(ref_path)
        gpri, gsec, _, gpos = hostlib.basecall(g["signal"], g["basecallpos"], 0.33)
        fwdp = orc.create_profile_trace(g["signal"], gpos, gpri, gsec, 0, 0)
        revp = orc.revcomp_profile(fwdp)
        trimmed = orc.create_profile_trace(tr, bcpos, pri, sec, trim[0], trim[1])
        forward = orc.gotoh_score_prof(trimmed, fwdp, 1, 0, SC) > orc.gotoh_score_prof(trimmed, revp, 1, 0, SC)
        refp = fwdp if forward else revp
        refslice = gpri if forward else so.revcomp(gpri)
        name, rpos = "wildtype", 0
        score, btr = orc.gotoh_prof(full, refp, 1, 0, SC)
    row0, row1 = orc.create_alignment_prof(btr, full, refp)
    padded = so.alignment_trace_padding(row0, tr, bcpos, pri, sec, con, q)
    return {
        ".txt": so.plot_alignment(row0, row1, name, rpos, len(refslice), forward, score, linelimit),
        ".json": so.trace_align_json(padded, name, rpos, forward, row0, row1),
        ".align.fa": so.align_fasta_text(stem, name, forward, row0, row1),
    }, (tr, pos)


def check_outputs(prefix, trace_path, ref_path, trim=(50, 50), linelimit=60):
    from tracy_amd import hostlib
    stem = os.path.splitext(os.path.basename(trace_path))[0]
    want, (tr, pos) = expected_files(trace_path, ref_path, stem, trim, linelimit)
    for ext, txt in want.items():
        assert open(prefix + ext).read() == txt, (prefix, ext)
    ref_txt = prefix + ".abif.expected"
    assert hostlib.trace_txt(ref_txt, tr, pos, 0.33, trim[0], trim[1]) == 0
    assert open(prefix + ".abif", "rb").read() == open(ref_txt, "rb").read()


@pytest.mark.parametrize("reverse", [False, True])
def test_config0_single_trace_vs_fasta(tmp_path, reverse):
    rng = np.random.default_rng(800 + reverse)
    trace_path, ref_path = synth_case(rng, str(tmp_path), "t%d" % reverse, reverse=reverse)
    prefix = str(tmp_path / "out")
    p = subprocess.run([CLI, "align", "-r", ref_path, "-o", prefix, trace_path], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert [ln.split("] ", 1)[1] for ln in p.stdout.strip().split("\n")][1:] == ["Load ab1 file", "Find reference match", "Alignment", "Output",
                                                                                 "Done."]
    check_outputs(prefix, trace_path, ref_path)
    head = open(prefix + ".txt").read().split("\n")
    assert head[0] == ">Alt" and ("reversecomplement" in open(prefix + ".txt").read()) == reverse


def test_wildtype_trace_reference_and_options(tmp_path):
    rng = np.random.default_rng(42)
    trace_path, ref_path = synth_case(rng, str(tmp_path), "w", nb=300, wildtype=True, reverse=True)
    prefix = str(tmp_path / "wt")
    p = subprocess.run([CLI, "align", "--reference", ref_path, "--outprefix=" + prefix, "-q", "20", "-u", "30", "-l", "40", trace_path],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    check_outputs(prefix, trace_path, ref_path, trim=(20, 30), linelimit=40)


def test_batch_manifest_and_errors(tmp_path):
    rng = np.random.default_rng(7)
    rows = []
    for i in range(5):
        t, r = synth_case(rng, str(tmp_path), "b%d" % i, nb=int(rng.integers(200, 600)), nref=int(rng.integers(1500, 4000)), reverse=bool(i % 2),
                          wildtype=(i == 3))
        rows.append((t, r, str(tmp_path / ("res%d" % i))))
    man = str(tmp_path / "manifest.tsv")
    open(man, "w").write("# trace\treference\tprefix\n" + "".join("\t".join(r) + "\n" for r in rows))
    p = subprocess.run([CLI, "align", "--batch", man], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    for t, r, pre in rows:
        check_outputs(pre, t, r)
    # error behaviour of the reference CLI: missing files -> exit 1, junk trace -> 255 (-1), trims too large -> 255
    p = subprocess.run([CLI, "align", "-r", str(tmp_path / "nope.fa"), rows[0][0]], capture_output=True, text=True)
    assert p.returncode == 1 and "Reference file is missing" in p.stderr
    junk = str(tmp_path / "junk.ab1")
    open(junk, "wb").write(b"hello world, not a trace")
    p = subprocess.run([CLI, "align", "-r", rows[0][1], junk], capture_output=True, text=True)
    assert p.returncode == 255 and "Unknown trace file type!" in p.stderr
    p = subprocess.run([CLI, "align", "-r", rows[0][1], "-q", "5000", "-o", str(tmp_path / "x"), rows[0][0]], capture_output=True, text=True)
    assert p.returncode == 255 and "larger than the trace" in p.stderr
    p = subprocess.run([CLI, "align"], capture_output=True, text=True)
    assert p.returncode == 255 and "Usage: tracy align" in p.stdout
