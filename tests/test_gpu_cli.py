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
    # output files that cannot be written (the prefix's directory does not exist) are reported and count in the exit code, also in --batch mode
    bad_rows = [rows[0][:2] + (str(tmp_path / "no_such_dir" / "res"),), rows[1]]
    man2 = str(tmp_path / "manifest2.tsv")
    open(man2, "w").write("".join("\t".join(r) + "\n" for r in bad_rows))
    p = subprocess.run([CLI, "align", "--batch", man2], capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "Could not write output file" in p.stderr and "no_such_dir" in p.stderr, (p.returncode, p.stderr[-500:])
    check_outputs(bad_rows[1][2], bad_rows[1][0], bad_rows[1][1])
    # an index written by an earlier format version is named as such
    old_idx = str(tmp_path / "old.tidx")
    open(old_idx, "wb").write(b"TAMDIDX1" + bytes(64))
    p = subprocess.run([CLI, "align", "-r", old_idx, "-o", str(tmp_path / "y"), rows[0][0]], capture_output=True, text=True)
    assert p.returncode != 0 and "older version" in p.stderr and "index" in p.stderr, p.stderr[-400:]


def test_batch_on_a_device_group(tmp_path):
    """-d 0,0: the manifest's traces are cut into one block per listed GPU (tracyhip_group_align_traces); same files"""
    rng = np.random.default_rng(9)
    cases = [synth_case(rng, str(tmp_path), "g%d" % i, nb=int(rng.integers(300, 700)), nref=int(rng.integers(1500, 3000)), reverse=bool(i % 2))
             for i in range(6)]
    outs = {}
    for tag, dev in (("one", "0"), ("two", "0,0")):
        rows = [(cases[i % 6][0], cases[i % 6][1], str(tmp_path / ("%s_%03d" % (tag, i)))) for i in range(140)]
        man = str(tmp_path / ("manifest_%s.tsv" % tag))
        open(man, "w").write("".join("\t".join(r) + "\n" for r in rows))
        p = subprocess.run([CLI, "align", "--batch", man, "-d", dev], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr
        outs[tag] = rows
    for (t, r, a), (_, _, b) in zip(outs["one"], outs["two"]):
        for ext in (".align.fa", ".txt", ".json"):
            assert open(a + ext, "rb").read() == open(b + ext, "rb").read(), (a, ext)
    check_outputs(outs["two"][139][2], outs["two"][139][0], outs["two"][139][1])
    p = subprocess.run([CLI, "align", "--batch", man, "-d", "0,x"], capture_output=True, text=True)
    assert p.returncode == 255


# ---- `decompose` ------------------------------------------------------------------------------------------
def decompose_case(tmp, tag, seed, n=1500, mf=500, kind=0, reverse=False):
    from tracy_amd import hostlib
    ref, sig, pos, indel = hostlib.synth_decompose(seed, n, mf, 12, kind, 0.6)
    trace_path = os.path.join(tmp, tag + ".ab1")
    hostlib.write_abif(trace_path, np.minimum(sig, 32000), pos, b"N" * len(pos), np.full(len(pos), 30, np.uint8))
    if reverse:
        ref = so.revcomp(ref)
    ref_path = os.path.join(tmp, tag + ".fa")
    open(ref_path, "w").write(">amplicon_%s\n%s\n" % (tag, ref.decode()))
    return trace_path, ref_path, indel


def expected_decompose(trace_path, ref_path, trim=(50, 50), linelimit=60, variants=True):
    import indigo_oracle as io
    from tracy_amd import hostlib
    t = hostlib.read_trace(trace_path)
    tr, pos = t["signal"], t["basecallpos"]
    pri, sec, con, bcpos, q = hostlib.basecall_qual(tr, pos, 0.33)
    name, ref = so.load_single_fasta(ref_path)
    w = io.decompose_trace(tr, bcpos, pri, sec, ref.encode(), SC, trim[0], trim[1])
    assert w["status"] == 0
    forward = bool(w["forward"])
    refslice = ref.encode() if forward else so.revcomp(ref.encode())
    p_t, s_t = io.trimmed_seq(w["primary"], *trim), io.trimmed_seq(w["secdecomp"], *trim)
    rep = dict(a1a2=w["af"], dcp=w["dcp"], indelshift=bool(w["bp"].indelshift), breakpoint=int(w["bp"].breakpoint), var=[])
    files = {".decomp": io.write_decomposition(w["dcp"])}
    slices = []
    for k, seq in enumerate((p_t, s_t)):
        sl = refslice[w["slice_begin%d" % k]:w["slice_begin%d" % k] + w["slice_len%d" % k]]
        slices.append(sl)
        rows = orc.create_alignment_str(w["btr%d" % k], seq, sl)
        rep["align%d" % (k + 1)] = rows
        rep["score%d" % (k + 1)] = w["score%d" % k]
        rep["rs%d" % (k + 1)] = dict(chr=name, pos=w["ref_pos%d" % k], forward=forward)
        files[".align%d" % (k + 1)] = so.plot_alignment(rows[0], rows[1], name, w["ref_pos%d" % k], len(sl), forward, w["score%d" % k], linelimit,
                                                         key=k + 1, a1a2=w["af"])
    rows3 = orc.create_alignment_str(w["btr2"], p_t, s_t)
    rep["align3"], rep["score3"] = rows3, w["score2"]
    files[".align3"] = so.plot_alignment(rows3[0], rows3[1], "Alt2", 0, len(s_t), True, w["score2"], linelimit, key=3, a1a2=w["af"])
    if not rep["indelshift"]:
        rep["breakpoint"] = io.nearest_snp(trim[0], trim[1], w["primary"], w["secondary"], so.find_best_trace_section(w["secondary"], bcpos.tolist())[1])
    if variants:
        for k, seq in enumerate((p_t, s_t)):
            rs = rep["rs%d" % (k + 1)]
            if forward:
                io.call_variants(rep["align%d" % (k + 1)][0], rep["align%d" % (k + 1)][1], name, rs["pos"], rep["var"])
            else:
                rseq, rsl = so.revcomp(seq), so.revcomp(slices[k])
                _, btr = orc.gotoh_str(rseq, rsl, 1, 0, SC)
                r0, r1 = orc.create_alignment_str(btr, rseq, rsl)
                io.call_variants(r0, r1, name, rs["pos"], rep["var"])
        io.sort_variants(rep["var"])
    cfg = dict(trimLeft=trim[0], trimRight=trim[1], pratio=0.33, genome=os.path.basename(ref_path), input=os.path.basename(trace_path), qualCut=45)
    files[".json"] = io.allele_json(cfg, tr, bcpos.tolist(), q.tolist(), w["primary"], w["secondary"], rep)
    return files, rep, w


@pytest.mark.parametrize("reverse", [False, True])
def test_decompose_cli_single_trace(tmp_path, reverse):
    trace_path, ref_path, indel = decompose_case(str(tmp_path), "d%d" % reverse, 4200 + reverse, reverse=reverse)
    prefix = str(tmp_path / "dec")
    p = subprocess.run([CLI, "decompose", "-v", "-r", ref_path, "-o", prefix, trace_path], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    files, rep, w = expected_decompose(trace_path, ref_path)
    for ext, txt in files.items():
        assert open(prefix + ext).read() == txt, ext
    assert rep["indelshift"] and indel != 0 and any(a == indel for a, b in rep["dcp"])
    assert len(rep["var"]) > 0
    recs = [ln.split("\t") for ln in open(prefix + ".vcf").read().split("\n") if ln and not ln.startswith("#")]
    assert [(r[0], int(r[1]), r[3], r[4]) for r in recs] == [(v["chr"], v["pos"], v["ref"], v["alt"]) for v in rep["var"]]
    assert [r[9].split(":")[0] for r in recs] == [{0: "0/0", 1: "0/1", 2: "1/1"}[v["gt"]] for v in rep["var"]]
    # ... and <prefix>.bcf, the file the reference writes (variants.h:141-261), decoded by the tests' own BCF2 reader
    from bcf_reader import read_bcf
    header, brecs, _ = read_bcf(prefix + ".bcf")
    assert header.endswith("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsample\n")
    assert [(b["CHROM"], b["POS"], b["REF"], b["ALT"], b["GT"]) for b in brecs] == [(r[0], int(r[1]), r[3], r[4], r[9].split(":")[0]) for r in recs]
    steps = [ln.split("] ", 1)[1] for ln in p.stdout.strip().split("\n") if ln.startswith("[")][1:]
    assert steps == ["Load ab1 file", "Find Reference Match", "Alignment", "InDel Search", "Decompose Chromatogram", "Estimate allelic fractions",
                     "Allele-specific alignments", "Variant Calling", "Done."]


def test_decompose_cli_batch_and_failures(tmp_path):
    rows = []
    for i in range(4):
        t, r, _ = decompose_case(str(tmp_path), "m%d" % i, 5100 + i, n=1200 + 100 * i, mf=400 + 30 * i, kind=(1 if i == 2 else 0), reverse=bool(i % 2))
        rows.append((t, r, str(tmp_path / ("dres%d" % i))))
    man = str(tmp_path / "manifest.tsv")
    open(man, "w").write("".join("\t".join(r) + "\n" for r in rows))
    p = subprocess.run([CLI, "decompose", "--batch", man, "-l", "50"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    for t, r, pre in rows:
        files, rep, w = expected_decompose(t, r, linelimit=50, variants=False)
        for ext, txt in files.items():
            assert open(pre + ext).read() == txt, (pre, ext)
    # a reference far shorter than the trace cannot reach the score gate of indigo.h:303-309
    rng = np.random.default_rng(1)
    bad = str(tmp_path / "unrelated.fa")
    open(bad, "w").write(">unrelated\n%s\n" % bytes(rng.choice(list(b"ACGT"), size=30).tolist()).decode())
    p = subprocess.run([CLI, "decompose", "-r", bad, "-o", str(tmp_path / "bad"), rows[0][0]], capture_output=True, text=True, timeout=300)
    assert p.returncode == 255 and "Alignment of trace to reference failed!" in p.stderr


# ---- indexed genome: host k-mer seeding + device extend (BASELINE configs[3]) ---------------------------------
def write_genome(rng, tmp, ncontigs=3, clen=20000):
    import gzip
    contigs = [("ctg%d" % i, bytes(rng.choice(list(b"ACGT"), size=clen + 777 * i).tolist()).decode()) for i in range(ncontigs)]
    path = os.path.join(tmp, "genome.fa.gz")
    with gzip.open(path, "wt") as f:
        for name, seq in contigs:
            f.write(">%s some description\n" % name)
            for i in range(0, len(seq), 80):
                f.write(seq[i:i + 80] + "\n")
    return path, contigs


def genomic_trace(rng, tmp, tag, contigs, nb=500, reverse=False):
    """ABIF trace whose bases are a stretch of a contig (with a few edits)"""
    from tracy_amd import hostlib
    ci = int(rng.integers(0, len(contigs)))
    seq = contigs[ci][1]
    start = int(rng.integers(0, len(seq) - nb - 10))
    core = bytearray(seq[start:start + nb + 6].encode())
    del core[200:203]
    core[250] = ord("A") if core[250] != ord("A") else ord("G")
    core = bytes(core[:nb])
    if reverse:
        core = so.revcomp(core)
    tr = np.zeros((4, 12 * nb + 12), np.int32)
    pos = 6 + 12 * np.arange(nb, dtype=np.int32)
    tri = 1.0 - np.abs(np.arange(-5, 6)) / 6.0
    for j, ch in enumerate(core):
        amp = rng.uniform(500, 1100)
        tr[b"ACGT".index(ch), pos[j] - 5:pos[j] + 6] += (amp * tri).astype(np.int32)
        tr[int(rng.integers(0, 4)), pos[j] - 5:pos[j] + 6] += (amp * 0.08 * tri).astype(np.int32)
    path = os.path.join(tmp, tag + ".ab1")
    hostlib.write_abif(path, tr, pos, core, np.full(nb, 40, np.uint8))
    return path


def expected_indexed(trace_path, contigs, stem, linelimit=60):
    from tracy_amd import hostlib
    t = hostlib.read_trace(trace_path)
    tr, pos = t["signal"], t["basecallpos"]
    pri, sec, con, bcpos, q = hostlib.basecall_qual(tr, pos, 0.33)
    full = orc.create_profile_trace(tr, bcpos, pri, sec, 0, 0)
    g = so.BruteGenome(contigs)
    rs = so.get_reference_slice(g, con.decode())
    assert rs is not None
    r = so.align_trace_oriented(full, rs["refslice"].encode(), rs["forward"], SC)
    refp = orc.create_profile_str(r["refslice"])
    row0, row1 = orc.create_alignment_prof(r["btr"], full, refp)
    padded = so.alignment_trace_padding(row0, tr, bcpos, pri, sec, con, q)
    rpos = rs["pos"] + r["ref_pos"]
    return {
        ".txt": so.plot_alignment(row0, row1, rs["chr"], rpos, len(r["refslice"]), rs["forward"], r["score_final"], linelimit),
        ".json": so.trace_align_json(padded, rs["chr"], rpos, rs["forward"], row0, row1),
        ".align.fa": so.align_fasta_text(stem, rs["chr"], rs["forward"], row0, row1),
    }, rs


def test_align_against_indexed_genome(tmp_path):
    rng = np.random.default_rng(2718)
    gpath, contigs = write_genome(rng, str(tmp_path))
    rows = []
    for i in range(4):
        t = genomic_trace(rng, str(tmp_path), "g%d" % i, contigs, nb=int(rng.integers(300, 700)), reverse=bool(i % 2))
        rows.append((t, gpath, str(tmp_path / ("gres%d" % i))))
    # single-trace form
    p = subprocess.run([CLI, "align", "-r", gpath, "-o", rows[0][2], rows[0][0]], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "Load FM-Index" in p.stdout
    want, rs = expected_indexed(rows[0][0], contigs, "g0")
    for ext, txt in want.items():
        assert open(rows[0][2] + ext).read() == txt, ext
    # batch form: one index, one device batch
    man = str(tmp_path / "gm.tsv")
    open(man, "w").write("".join("\t".join(r) + "\n" for r in rows))
    p = subprocess.run([CLI, "align", "--batch", man], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    seen = set()
    for t, _, pre in rows:
        want, rs = expected_indexed(t, contigs, os.path.splitext(os.path.basename(t))[0])
        seen.add(rs["forward"])
        for ext, txt in want.items():
            assert open(pre + ext).read() == txt, (pre, ext)
    assert seen == {True, False}
    # a trace that is nowhere in the genome cannot be anchored
    lost = genomic_trace(rng, str(tmp_path), "lost", [("x", bytes(rng.choice(list(b"ACGT"), size=5000).tolist()).decode())])
    p = subprocess.run([CLI, "align", "-r", gpath, "-o", str(tmp_path / "lost"), lost], capture_output=True, text=True, timeout=300)
    assert p.returncode == 255 and "Couldn't anchor the Sanger trace" in p.stderr


def test_oriented_pipeline_mode_through_the_abi():
    """tracyhip_align_traces with job.oriented (no orientation scores) == the oracle's indexed-genome branch"""
    import tracy_amd
    from tracy_amd import hostlib
    refs, profs, rev = hostlib.synth_align(4400, 12, 2600, 500, 2)
    ctx = tracy_amd.Context(0)
    windows, fwd = [], []
    for i in range(12):
        w = refs[i].tobytes()
        if rev[i]:
            w = so.revcomp(w)          # what seeding hands over: the window in trace orientation
        windows.append(w)
        fwd.append(0 if rev[i] else 1)
    got = ctx.align_traces(list(profs), windows, SC, 50, 50, oriented=fwd)
    for i in range(12):
        want = so.align_trace_oriented(profs[i], windows[i], bool(fwd[i]), SC)
        for k in ("score_prelim", "slice_begin", "slice_len", "ref_pos", "score_final"):
            assert int(got[k][i]) == int(want[k]), (i, k)
        assert got["btr"][i] == want["btr"] and int(got["forward"][i]) == fwd[i]
        assert int(got["score_fwd"][i]) == int(got["score_rev"][i]) == int(want["score_prelim"])
    ctx.close()


def test_decompose_against_indexed_genome(tmp_path):
    """indigo.h:213-218: seeding picks the window and the strand, the device chain runs without orientation scores"""
    import indigo_oracle as io
    from tracy_amd import hostlib
    rng = np.random.default_rng(31415)
    # the genome holds the reference allele of two synthetic heterozygous traces (one on each strand)
    cases = []
    contigs = []
    for i in range(2):
        ref, sig, pos, indel = hostlib.synth_decompose(8800 + i, 2600, 600, 10, 0, 0.6)
        body = ref.decode()
        flank = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).tolist()).decode()
        seq = flank(3000) + body + flank(2500)
        if i == 1:
            seq = so.revcomp(seq.encode()).decode()
        contigs.append(("chr%d" % (i + 1), seq))
        tpath = str(tmp_path / ("ix%d.ab1" % i))
        hostlib.write_abif(tpath, np.minimum(sig, 32000), pos, b"N" * len(pos), np.full(len(pos), 30, np.uint8))
        cases.append(tpath)
    import gzip
    gpath = str(tmp_path / "g.fa.gz")
    with gzip.open(gpath, "wt") as f:
        for name, seq in contigs:
            f.write(">%s\n%s\n" % (name, seq))
    man = str(tmp_path / "ix.tsv")
    open(man, "w").write("".join("%s\t%s\t%s\n" % (t, gpath, str(tmp_path / ("ixres%d" % i))) for i, t in enumerate(cases)))
    p = subprocess.run([CLI, "decompose", "--batch", man], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    g = so.BruteGenome(contigs)
    strands = set()
    for i, tpath in enumerate(cases):
        t = hostlib.read_trace(tpath)
        tr, pos = t["signal"], t["basecallpos"]
        pri, sec, con, bcpos, q = hostlib.basecall_qual(tr, pos, 0.33)
        rs = so.get_reference_slice(g, con.decode())
        assert rs is not None
        strands.add(rs["forward"])
        window = rs["refslice"].encode()
        w = io.decompose_trace(tr, bcpos, pri, sec, window, SC, 50, 50, oriented_forward=rs["forward"])
        assert w["status"] == 0
        pre = str(tmp_path / ("ixres%d" % i))
        assert open(pre + ".decomp").read() == io.write_decomposition(w["dcp"])
        p_t, s_t = io.trimmed_seq(w["primary"], 50, 50), io.trimmed_seq(w["secdecomp"], 50, 50)
        for k, seq in enumerate((p_t, s_t)):
            sl = window[w["slice_begin%d" % k]:w["slice_begin%d" % k] + w["slice_len%d" % k]]
            rows = orc.create_alignment_str(w["btr%d" % k], seq, sl)
            want = so.plot_alignment(rows[0], rows[1], rs["chr"], rs["pos"] + w["ref_pos%d" % k], len(sl), rs["forward"], w["score%d" % k], 60,
                                     key=k + 1, a1a2=w["af"])
            assert open(pre + ".align%d" % (k + 1)).read() == want, (i, k)
    assert strands == {True, False}


def test_decompose_against_wildtype_trace(tmp_path):
    """indigo.h:249-289: the reference is a wildtype chromatogram -- profile x profile preliminary alignment, allele
    alignments against its primary basecalls"""
    import indigo_oracle as io
    from tracy_amd import hostlib
    for reverse in (False, True):
        ref, sig, pos, indel = hostlib.synth_decompose(9900 + reverse, 700, 520, 8, 0, 0.6)
        tpath = str(tmp_path / ("wtd%d.ab1" % reverse))
        hostlib.write_abif(tpath, np.minimum(sig, 32000), pos, b"N" * len(pos), np.full(len(pos), 30, np.uint8))
        # wildtype trace: clean peaks of the reference window (reverse strand for the second case)
        wseq = ref if not reverse else so.revcomp(ref)
        wt = np.zeros((4, 12 * len(wseq) + 12), np.int32)
        wpos = 6 + 12 * np.arange(len(wseq), dtype=np.int32)
        tri = (900 * (1.0 - np.abs(np.arange(-5, 6)) / 6.0)).astype(np.int32)
        for j, ch in enumerate(wseq):
            wt[b"ACGT".index(ch), wpos[j] - 5:wpos[j] + 6] += tri
        wpath = str(tmp_path / ("wtref%d.ab1" % reverse))
        hostlib.write_abif(wpath, wt, wpos, wseq, np.full(len(wseq), 40, np.uint8))
        pre = str(tmp_path / ("wtres%d" % reverse))
        p = subprocess.run([CLI, "decompose", "-r", wpath, "-o", pre, tpath], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr
        t = hostlib.read_trace(tpath)
        tr, tpos = t["signal"], t["basecallpos"]
        pri, sec, con, bcpos, q = hostlib.basecall_qual(tr, tpos, 0.33)
        g = hostlib.read_trace(wpath)
        gpri, gsec, _, gpos = hostlib.basecall(g["signal"], g["basecallpos"], 0.33)
        wprof = orc.create_profile_trace(g["signal"], gpos, gpri, gsec, 0, 0)
        w = io.decompose_trace(tr, bcpos, pri, sec, gpri, SC, 50, 50, wildtype_profile=wprof)
        assert w["status"] == 0 and bool(w["forward"]) == (not reverse)
        refslice = gpri if w["forward"] else so.revcomp(gpri)
        assert open(pre + ".decomp").read() == io.write_decomposition(w["dcp"])
        p_t, s_t = io.trimmed_seq(w["primary"], 50, 50), io.trimmed_seq(w["secdecomp"], 50, 50)
        for k, seq in enumerate((p_t, s_t)):
            sl = refslice[w["slice_begin%d" % k]:w["slice_begin%d" % k] + w["slice_len%d" % k]]
            rows = orc.create_alignment_str(w["btr%d" % k], seq, sl)
            want = so.plot_alignment(rows[0], rows[1], "wildtype", w["ref_pos%d" % k], len(sl), bool(w["forward"]), w["score%d" % k], 60,
                                     key=k + 1, a1a2=w["af"])
            assert open(pre + ".align%d" % (k + 1)).read() == want, (reverse, k)


# ---- `assemble` (BASELINE configs[4] at test size) ---------------------------------------------------------------
def tiled_traces(rng, tmp, n, region_len=1500, tlen=420, some_reverse=True, noisy_ends=True):
    from tracy_amd import hostlib
    region = bytes(rng.choice(list(b"ACGT"), size=region_len).tolist())
    paths = []
    for i in range(n):
        start = int(i * (region_len - tlen) / max(n - 1, 1))
        seq = bytearray(region[start:start + tlen])
        for k in range(len(seq)):
            if rng.random() < 0.01:
                seq[k] = int(rng.choice(list(b"ACGT")))
        seq = bytes(seq)
        if some_reverse and i % 3 == 1:
            seq = so.revcomp(seq)
        nb = len(seq)
        tr = np.zeros((4, 12 * nb + 12), np.int32)
        pos = 6 + 12 * np.arange(nb, dtype=np.int32)
        tri = 1.0 - np.abs(np.arange(-5, 6)) / 6.0
        for j, ch in enumerate(seq):
            amp = rng.uniform(500, 1100)
            tr[b"ACGT".index(ch), pos[j] - 5:pos[j] + 6] += (amp * tri).astype(np.int32)
            noisy = noisy_ends and (j < 25 or j > nb - 30)
            tr[int(rng.integers(0, 4)), pos[j] - 5:pos[j] + 6] += (amp * (0.6 if noisy else 0.06) * tri).astype(np.int32)
        p = os.path.join(tmp, "tile%02d.ab1" % i)
        hostlib.write_abif(p, tr, pos, seq, np.full(nb, 40, np.uint8))
        paths.append(p)
    return region, paths


def test_assemble_reference_guided(tmp_path):
    import assemble_oracle as ao
    rng = np.random.default_rng(1618)
    region, paths = tiled_traces(rng, str(tmp_path), 5)
    junk_region, junk = tiled_traces(np.random.default_rng(3), str(tmp_path / ".."), 1, region_len=500, tlen=300)
    ref_path = str(tmp_path / "region.fa")
    open(ref_path, "w").write(">region\n%s\n" % region.decode())
    pre = str(tmp_path / "asm")
    p = subprocess.run([CLI, "assemble", "-r", ref_path, "-o", pre, "-i", "-a", "fastq"] + paths + junk, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr
    assert "is not matching to the reference" in p.stderr
    want, sidx = ao.assemble_ref_guided(paths + junk, ref_path, SC, inccons=True, fmt="fastq")
    assert len(sidx) == 5 and not all(s["forward"] for s in sidx)
    for ext, txt in want.items():
        assert open(pre + ext).read() == txt, ext


def test_assemble_de_novo(tmp_path):
    import assemble_oracle as ao
    rng = np.random.default_rng(2236)
    region, paths = tiled_traces(rng, str(tmp_path), 5, region_len=1200, tlen=400)
    _, lonely = tiled_traces(np.random.default_rng(8), str(tmp_path / ".."), 1, region_len=500, tlen=300)
    pre = str(tmp_path / "dn")
    p = subprocess.run([CLI, "assemble", "-o", pre, "--called", "0.3"] + paths + lonely, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr
    assert "is not matching to any of the other traces" in p.stderr
    want, info = ao.assemble_denovo(paths + lonely, SC, called=0.3)
    assert info["keep"] == [0, 1, 2, 3, 4]
    for ext, txt in want.items():
        assert open(pre + ext).read() == txt, ext
    cons = open(pre + ".cons.fa").read().split("\n")[1]
    assert len(cons) > 1000  # the tiles were merged into one contig of about the region's length


# ---- `consensus` ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("reverse,opts", [(False, []), (True, ["-a", "-i", "-b", "amplicon7", "-l", "45", "-q", "30", "-s", "20"])])
def test_consensus_of_two_traces(tmp_path, reverse, opts):
    import consensus_oracle as co
    rng = np.random.default_rng(4242 + reverse)
    region, paths = tiled_traces(rng, str(tmp_path), 2, region_len=700, tlen=480, some_reverse=False, noisy_ends=False)
    if reverse:  # second trace from the other strand
        _, alt = tiled_traces(np.random.default_rng(4243), str(tmp_path / ".."), 2, region_len=700, tlen=480, some_reverse=False, noisy_ends=False)
        from tracy_amd import hostlib
        t = hostlib.read_trace(paths[1])
        sig = np.ascontiguousarray(t["signal"][::-1, ::-1])
        pos = (sig.shape[1] - 1 - t["basecallpos"][::-1]).astype(np.int32)
        hostlib.write_abif(paths[1], sig, pos, b"N" * len(pos), np.full(len(pos), 30, np.uint8))
    pre = str(tmp_path / "cons")
    p = subprocess.run([CLI, "consensus", "-o", pre] + opts + paths, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    kw = {}
    if opts:
        kw = dict(trims=(30, 50, 50, 20), label="amplicon7", union=False, iupac=True, linelimit=45)
    want, forward = co.consensus(paths[0], paths[1], SC, **kw)
    assert forward == (not reverse)
    for ext, txt in want.items():
        assert open(pre + ext).read() == txt, ext
    assert os.path.exists(pre + "_1st.abif") and os.path.exists(pre + "_2nd.abif")


def test_consensus_without_overlap(tmp_path):
    rng = np.random.default_rng(5)
    _, a = tiled_traces(rng, str(tmp_path), 1, region_len=400, tlen=300, some_reverse=False)
    os.rename(a[0], str(tmp_path / "x.ab1"))
    _, b = tiled_traces(np.random.default_rng(6), str(tmp_path), 1, region_len=400, tlen=300, some_reverse=False)
    p = subprocess.run([CLI, "consensus", "-o", str(tmp_path / "n"), str(tmp_path / "x.ab1"), b[0]], capture_output=True, text=True, timeout=600)
    assert p.returncode == 1 and "No sufficient trace overlap" in p.stderr
    p = subprocess.run([CLI, "consensus", b[0]], capture_output=True, text=True)
    assert p.returncode == 1 and "Exactly 2 input trace files" in p.stderr
