"""Host writers of `tracy decompose` (tracy_amd/host/indigo_out.hpp: .decomp, .align1-3, .json, variants)
against the independent Python restatement in tests/indigo_oracle.py, on decompositions computed by the oracle
chain (CPU only).  PARITY UNPINNED (json.h / variants.h need htslib + Boost)."""
import os

import numpy as np
import pytest

import indigo_oracle as io
import pyoracle as orc
import sage_oracle as so

SC = (3, -5, -10, -4)


@pytest.mark.parametrize("reverse,kind", [(False, 0), (True, 0), (False, 1)])
def test_decompose_output_files(tmp_path, reverse, kind):
    from tracy_amd import hostlib
    ref, sig, pos, indel = hostlib.synth_decompose(900 + 2 * kind + reverse, 900, 300, 9, kind, 0.6)
    if reverse:
        ref = so.revcomp(ref)
    trim = (30, 40)
    pri, sec, con, bcpos, q = hostlib.basecall_qual(sig, pos, 0.33)
    w = io.decompose_trace(sig, bcpos, pri, sec, ref, SC, trim[0], trim[1])
    assert w["status"] == 0
    forward = bool(w["forward"])
    refslice = ref if forward else so.revcomp(ref)
    p_t, s_t = io.trimmed_seq(w["primary"], *trim), io.trimmed_seq(w["secdecomp"], *trim)
    rows, var_rows, slices, var = [], [], [], []
    for k, seq in enumerate((p_t, s_t)):
        sl = refslice[w["slice_begin%d" % k]:w["slice_begin%d" % k] + w["slice_len%d" % k]]
        slices.append(sl)
        rows.append(orc.create_alignment_str(w["btr%d" % k], seq, sl))
        if forward:
            var_rows.append(rows[k])
        else:
            rseq, rsl = so.revcomp(seq), so.revcomp(sl)
            var_rows.append(orc.create_alignment_str(orc.gotoh_str(rseq, rsl, 1, 0, SC)[1], rseq, rsl))
        io.call_variants(var_rows[k][0], var_rows[k][1], "chrZ", w["ref_pos%d" % k], var)
    io.sort_variants(var)
    rows.append(orc.create_alignment_str(w["btr2"], p_t, s_t))
    prefix = str(tmp_path / "d")
    rc = hostlib.decompose_outputs(prefix, "ref.fa", "trace.ab1", sig, pos, 0.33, trim, 45, 60, w["primary"], w["secondary"], w["secdecomp"], rows,
                                   var_rows, "chrZ", forward, (w["ref_pos0"], w["ref_pos1"]), (len(slices[0]), len(slices[1])), len(refslice),
                                   (w["score0"], w["score1"], w["score2"]), bool(w["bp"].indelshift), int(w["bp"].breakpoint), w["af"], w["dcp"])
    assert rc == 0
    bpnt = int(w["bp"].breakpoint)
    if not w["bp"].indelshift:
        bpnt = io.nearest_snp(trim[0], trim[1], w["primary"], w["secondary"], so.find_best_trace_section(w["secondary"], bcpos.tolist())[1])
    rep = dict(a1a2=w["af"], dcp=w["dcp"], indelshift=bool(w["bp"].indelshift), breakpoint=bpnt, var=var, align1=rows[0], align2=rows[1],
               align3=rows[2], score1=w["score0"], score2=w["score1"], score3=w["score2"],
               rs1=dict(chr="chrZ", pos=w["ref_pos0"], forward=forward), rs2=dict(chr="chrZ", pos=w["ref_pos1"], forward=forward))
    cfg = dict(trimLeft=trim[0], trimRight=trim[1], pratio=0.33, genome="ref.fa", input="trace.ab1", qualCut=45)
    assert open(prefix + ".json").read() == io.allele_json(cfg, sig, bcpos.tolist(), q.tolist(), w["primary"], w["secondary"], rep)
    assert open(prefix + ".decomp").read() == io.write_decomposition(w["dcp"])
    for k in range(2):
        assert open(prefix + ".align%d" % (k + 1)).read() == so.plot_alignment(rows[k][0], rows[k][1], "chrZ", w["ref_pos%d" % k], len(slices[k]), forward,
                                                                              w["score%d" % k], 60, key=k + 1, a1a2=w["af"])
    assert open(prefix + ".align3").read() == so.plot_alignment(rows[2][0], rows[2][1], "Alt2", 0, len(s_t), True, w["score2"], 60, key=3, a1a2=w["af"])
    recs = [ln.split("\t") for ln in open(prefix + ".vcf").read().split("\n") if ln and not ln.startswith("#")]
    assert [(r[0], int(r[1]), r[3], r[4]) for r in recs] == [(v["chr"], v["pos"], v["ref"], v["alt"]) for v in var]
    # <prefix>.bcf (vcfOutput, variants.h:141-261: BGZF + BCF2.2 written without htslib) decoded by the tests' own reader: the header text
    # and every column of every record are those of the VCF text
    from bcf_reader import read_bcf
    header, brecs, blocks = read_bcf(prefix + ".bcf")
    vcf_lines = open(prefix + ".vcf").read().split("\n")
    # (the BCF header carries htslib's IDX keys: the dictionary index of every FILTER / INFO / FORMAT / contig line, PASS = 0)
    import re
    assert re.sub(r",IDX=\d+>", ">", header) == "\n".join(ln for ln in vcf_lines if ln.startswith("#")) + "\n"
    idx = [(ln.split("<ID=")[1].split(",")[0], int(re.search(r",IDX=(\d+)>$", ln).group(1))) for ln in header.split("\n") if re.match(r"##(FILTER|INFO|FORMAT)=<", ln)]
    assert idx == [("PASS", 0), ("LowQual", 1), ("BASEPOS", 2), ("SIGNALPOS", 3), ("TYPE", 4), ("METHOD", 5), ("GT", 6), ("GQ", 7)]
    assert [int(re.search(r",IDX=(\d+)>$", ln).group(1)) for ln in header.split("\n") if ln.startswith("##contig=<")] == list(range(sum(ln.startswith("##contig=<") for ln in header.split("\n"))))
    assert blocks >= 2 and len(brecs) == len(recs)
    for b, r in zip(brecs, recs):
        info = dict(kv.split("=", 1) for kv in r[7].split(";"))
        assert (b["CHROM"], b["POS"], b["ID"] or ".", b["REF"], b["ALT"], b["FILTER"]) == (r[0], int(r[1]), r[2], r[3], r[4], r[6])  # (an id of "." is stored as missing)
        assert b["QUAL"] == float(r[5])
        assert (b["INFO"]["TYPE"], b["INFO"]["METHOD"], b["INFO"]["BASEPOS"], b["INFO"]["SIGNALPOS"]) == (info["TYPE"], info["METHOD"], int(info["BASEPOS"]), int(info["SIGNALPOS"]))
        assert r[8] == "GT:GQ" and (b["GT"], b["GQ"]) == (r[9].split(":")[0], int(r[9].split(":")[1]))
    # <prefix>.bcf.csi (bcf_index_build, variants.h:263): every record is found through the index at its own position, nothing where there
    # is nothing, the pseudo-bin counts the records (tests/bcf_reader.py reads the index from the specification's layout)
    from bcf_reader import csi_query, read_csi
    csi = read_csi(prefix + ".bcf.csi")
    assert csi["min_shift"] == 14 and len(csi["refs"]) == 1 and csi["n_no_coor"] == 0
    meta = ((1 << ((csi["depth"] + 1) * 3)) - 1) // 7 + 1
    if brecs:
        assert csi["refs"][0][meta][1][1] == (len(brecs), 0)
        for b in brecs:
            hit = csi_query(prefix + ".bcf", csi, 0, b["POS"] - 1, b["POS"] - 1 + len(b["REF"]))
            assert (b["POS"] - 1, len(b["REF"])) in hit
        lo = min(b["POS"] for b in brecs) - 1
        assert csi_query(prefix + ".bcf", csi, 0, 0, max(lo, 0)) == [] or lo == 0
        allr = csi_query(prefix + ".bcf", csi, 0, 0, 1 << 29)
        assert allr == sorted(set((b["POS"] - 1, len(b["REF"])) for b in brecs))
    if kind == 0:
        assert w["bp"].indelshift and len(var) > 0


def test_call_variants_known_answer():
    """hand-checked (rs.pos = 100, the two leading reference bases are skipped): SNV G>T at 105, deletion of GT
    anchored on the A at 106, insertion of AA anchored on the T at 111"""
    var = []
    row0 = b"--ACTA--CGTAAC"
    row1 = b"TTACGAGTCGT--C"
    io.call_variants(row0, row1, "c", 100, var)
    got = [(v["pos"], v["ref"], v["alt"], v["basenum"]) for v in var]
    assert got == [(105, "G", "T", 3), (106, "AGT", "A", 4), (111, "T", "TAA", 9)]
