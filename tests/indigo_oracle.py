"""The `tracy decompose` hot section (indigo.h:190-388, FASTA reference) composed from oracle functions --
the expected behaviour of tracyhip_decompose_traces (tests only)."""
import numpy as np

import pyoracle as orc
from sage_oracle import revcomp


def trimmed_seq(s, ltrim, rtrim):
    if ltrim + rtrim + 1 >= len(s):
        return s
    return s[ltrim:len(s) - rtrim]


def decompose_trace(sig, bcpos, pri, sec, ref, score, tl=50, tr=50, maxindel=1000, madc=5, oriented_forward=None, wildtype_profile=None):
    """oriented_forward: None = single-FASTA path (orientation by score, indigo.h:219-247); True/False = indexed-genome
    path (indigo.h:213-218): `ref` is the window already oriented by k-mer seeding and the flag is rs.forward"""
    trimmed = orc.create_profile_trace(sig, bcpos, pri, sec, tl, tr)
    bp = orc.find_breakpoint(trimmed)
    # wildtype_profile: the reference is a wildtype trace (indigo.h:249-289): `ref` = its primary basecalls, the
    # preliminary alignment runs against its profile
    fwdp = orc.create_profile_str(ref) if wildtype_profile is None else wildtype_profile
    if oriented_forward is None:
        revp = orc.revcomp_profile(fwdp)
        gs_fwd = orc.gotoh_score_prof(trimmed, fwdp, 1, 0, score)
        gs_rev = orc.gotoh_score_prof(trimmed, revp, 1, 0, score)
        forward = gs_fwd > gs_rev
        refslice = ref if forward else revcomp(ref)
        pref = fwdp if forward else revp
    else:
        gs_fwd = gs_rev = 0
        forward, refslice, pref = bool(oriented_forward), ref, fwdp
    sc1, btr1 = orc.gotoh_prof(trimmed, pref, 1, 0, score)
    rows = orc.create_alignment_prof(btr1, trimmed, pref)
    seqsize = float(trimmed.shape[1])
    status = 0
    if sc1 <= seqsize * 0.35 * score[0] + seqsize * (1 - 0.35) * score[1]:
        status = -1
    if not bp.indelshift:
        rc, bp = orc.find_homozygous_breakpoint(rows[0], rows[1], bp)
        if rc != 1 and status == 0:
            status = -2 if rc == 0 else -3
    p2, s2, dcp, st = orc.decompose_alleles(rows[0], rows[1], pri, sec, bp, len(refslice), tl, tr, maxindel, madc)
    sd = orc.generate_secondary_decomposed(sig, bcpos, p2, s2)
    af = orc.allelic_fraction(sig, bcpos, p2, sd, tl, tr)
    out = dict(bp=bp, status=status, score_fwd=gs_fwd, score_rev=gs_rev, forward=int(forward), score_trim=sc1, primary=p2,
               secondary=s2, dcp=dcp, dstatus=st, secdecomp=sd, af=af)
    pri_t, sec_t = trimmed_seq(p2, tl, tr), trimmed_seq(sd, tl, tr)
    for k, seq in enumerate((pri_t, sec_t)):
        s, btr = orc.gotoh_str(seq, refslice, 1, 0, score)
        r0, r1 = orc.create_alignment_str(btr, seq, refslice)
        ri, risize, pos, _ = orc.trim_reference_slice(r0, r1, tl, tr, len(refslice), forward)
        sl = refslice[ri:ri + risize]
        s2k, btr2 = orc.gotoh_str(seq, sl, 1, 0, score)
        out["slice_begin%d" % k] = ri
        out["slice_len%d" % k] = len(sl)
        out["ref_pos%d" % k] = pos
        out["score%d" % k] = s2k
        out["btr%d" % k] = btr2
    s3, btr3 = orc.gotoh_str(pri_t, sec_t, 0, 0, score)
    out["score2"] = s3
    out["btr2"] = btr3
    return out


# ---------------------------------------------------------------------------------------------------------
# `tracy decompose` writers restated in Python from the reference text (tests only; cross-checks
# tracy_amd/host/indigo_out.hpp).  PARITY UNPINNED (json.h / variants.h need htslib + Boost).
# ---------------------------------------------------------------------------------------------------------
def write_decomposition(dcp):
    """decompose.h:622-632"""
    return "indel\tdecomp\n" + "".join("%d\t%d\n" % (a, b) for a, b in dcp)


def call_variants(row0, row1, chrom, pos, var):
    """variants.h:34-126; var: list of dicts(pos, basenum, gt, chr, ref, alt, id), updated in place"""
    r0, r1 = row0.decode(), row1.decode()

    def insert(p, bnum, gt, ref, alt):
        for v in var:
            if v["pos"] == p and v["chr"] == chrom and v["ref"] == ref and v["alt"] == alt:
                v["gt"] += 1
                return
        if p > 0 and "n" not in ref.lower():
            var.append(dict(pos=p, basenum=bnum, gt=gt, chr=chrom, ref=ref, alt=alt, id="."))
    ri, vi_start, vi_end = pos, -1, -1
    for j in range(len(r0)):
        if r0[j] != "-":
            if vi_start == -1:
                vi_start = j
            vi_end = j
        if r1[j] != "-" and vi_start == -1:
            ri += 1
    vi, dele, ins, del_start, ins_start, last_ref = 0, "", "", 0, 0, "N"
    j = vi_start
    while vi_start >= 0 and j <= vi_end:
        if dele and r0[j] != "-":
            insert(del_start, vi, 1, dele, dele[0])
            dele = ""
        if ins and r1[j] != "-":
            insert(ins_start, vi, 1, ins[0], ins)
            ins = ""
        if r0[j] != "-":
            vi += 1
        if r1[j] != "-":
            ri += 1
        if r0[j] != r1[j]:
            if r0[j] != "-" and r1[j] != "-":
                insert(ri, vi, 1, r1[j], r0[j])
            elif r0[j] == "-":
                if not dele:
                    dele = last_ref
                    del_start = ri - 1
                dele += r1[j]
            else:
                if not ins:
                    ins = last_ref
                    ins_start = ri
                ins += r0[j]
        if r1[j] != "-":
            last_ref = r1[j]
        j += 1


def variant_type(ref, alt):
    if len(ref) == 1 and len(alt) == 1:
        return "SNV"
    return "Deletion" if len(ref) > len(alt) else "Insertion" if len(ref) < len(alt) else "Complex"


def x_window(bcpos, p):
    lb = bcpos[p] + 1
    lb = 1 if lb <= 150 else lb - 150
    ub = bcpos[p] + 1
    ub = ub + 150 if ub + 150 < bcpos[-1] else bcpos[-1]
    return lb, ub


_EXPAND = {"A": "A", "C": "C", "G": "G", "T": "T", "N": "N", "R": "A|G", "Y": "C|T", "S": "C|G", "W": "A|T", "K": "G|T", "M": "A|C"}


def nearest_snp(trim_left, trim_right, primary, secondary, rtp):
    """trim.h:10-33"""
    offset = 0
    while True:
        dead = True
        if rtp + offset + trim_right < len(secondary) and rtp + offset + trim_right < len(primary):
            if trim_left < rtp + offset and primary[rtp + offset] != secondary[rtp + offset]:
                return rtp + offset - trim_left
            dead = False
        if offset + trim_left < rtp:
            if primary[rtp - offset] != secondary[rtp - offset]:
                return rtp - offset - trim_left
            dead = False
        offset += 1
        if dead:
            break
    return rtp - trim_left if rtp > trim_left else trim_left


def trace_json_body(signal, bcpos, estqual, primary, secondary):
    """_traceJsonOut, json.h:33-105"""
    ns = len(signal[0])
    pri, sec = primary.decode(), secondary.decode()
    o = []
    o.append("\"pos\": [%s],\n" % ", ".join(str(i + 1) for i in range(ns)))
    for k, nm in enumerate(("peakA", "peakC", "peakG", "peakT")):
        o.append("\"%s\": [%s],\n" % (nm, ", ".join(str(int(v)) for v in signal[k])))
    vis, bc, idx = [], 0, bcpos[0]
    for i in range(ns):
        if idx == i:
            vis.append((i, bc))
            if bc < len(bcpos) - 1:
                bc += 1
                idx = bcpos[bc]

    def joined(items):
        return "".join((", " if i != bcpos[0] else "") + txt for (i, txt) in items)
    o.append("\"basecallPos\": [%s],\n" % joined([(i, str(i + 1)) for i, _ in vis]))
    o.append("\"basecallQual\": [%s],\n" % joined([(i, str(int(estqual[b]))) for i, b in vis]))
    items = []
    for i, b in vis:
        t = "\"%d\":\"%d:%s" % (i + 1, b + 1, pri[b])
        if pri[b] != sec[b]:
            t += "|" + _EXPAND.get(sec[b], "N")
        items.append((i, t + "\""))
    o.append("\"basecalls\": {%s},\n" % joined(items))
    o.append("\"primarySeq\": \"%s\",\n\"secondarySeq\": \"%s\"\n" % (pri, sec))
    return "".join(o)


def allele_json(cfg, signal, bcpos, estqual, primary, secondary, rep):
    """json.h:17-105 + 260-381.  cfg: dict(trimLeft, trimRight, pratio, genome, input, qualCut); rep: dict(rs1, rs2 (chr,pos,forward),
    align1..3 (row0,row1), score1..3, indelshift, breakpoint, a1a2, dcp, var)"""
    ns = len(signal[0])
    pri, sec = primary.decode(), secondary.decode()
    o = ["{\n"]
    o.append("\"meta\": {\"program\": \"tracy\", \"version\": \"0.9.1\", \"arguments\": {\"trimLeft\": %d, \"trimRight\": %d, \"pratio\": %s, "
             "\"genome\": \"%s\", \"input\": \"%s\"}},\n" % (cfg["trimLeft"], cfg["trimRight"], "%g" % cfg["pratio"], cfg["genome"], cfg["input"]))
    o.append(trace_json_body(signal, bcpos, estqual, primary, secondary))
    o.append(",\n")
    xw = x_window(bcpos, cfg["trimLeft"] + rep["breakpoint"])
    o.append("\"chartConfig\": { \"x\": { \"axis\": { \"range\": [%d, %d] }}},\n" % xw)
    for n in ("1", "2"):
        rs, al = rep["rs" + n], rep["align" + n]
        o.append("\"ref%schr\": \"%s\",\n\"ref%spos\": %d,\n\"alt%salign\": \"%s\",\n\"ref%salign\": \"%s\",\n\"ref%sforward\": %d,\n\"align%sscore\": %d,\n"
                 % (n, rs["chr"], n, rs["pos"] + 1, n, al[0].decode(), n, al[1].decode(), n, int(rs["forward"]), n, rep["score" + n]))
    o.append("\"allele1fraction\": %s,\n\"allele1align\": \"%s\",\n\"allele2fraction\": %s,\n\"allele2align\": \"%s\",\n\"align3score\": %d,\n"
             % ("%g" % rep["a1a2"][0], rep["align3"][0].decode(), "%g" % rep["a1a2"][1], rep["align3"][1].decode(), rep["score3"]))
    o.append("\"hetindel\": %d,\n" % int(rep["indelshift"]))
    o.append("\"decomposition\": {\n\"x\": [%s],\n\"y\": [%s]\n},\n" % (", ".join(str(a) for a, _ in rep["dcp"]), ", ".join(str(b) for _, b in rep["dcp"])))
    o.append("\"variants\": {\n\"columns\": [\"chr\", \"pos\", \"id\", \"ref\", \"alt\", \"qual\", \"filter\", \"type\", \"genotype\", \"basepos\", \"signalpos\"],\n")
    fwd = rep["rs1"]["forward"]
    rows, xr = [], []
    for v in rep["var"]:
        q = cfg["trimLeft"] + v["basenum"] - 1 if fwd else len(pri) - (cfg["trimRight"] + v["basenum"])
        gt = {0: "hom. REF", 1: "het.", 2: "hom. ALT"}.get(v["gt"], "missing")
        basepos = cfg["trimLeft"] + v["basenum"] if fwd else len(pri) - (cfg["trimRight"] + v["basenum"]) + 1
        rows.append("[\"%s\", %d, \"%s\", \"%s\", \"%s\", %d, \"%s\", \"%s\", \"%s\", %d, %d]"
                    % (v["chr"], v["pos"], v["id"], v["ref"], v["alt"], estqual[q], "LowQual" if estqual[q] < cfg["qualCut"] else "PASS",
                       variant_type(v["ref"], v["alt"]), gt, basepos, bcpos[q] + 1))
        xr.append("[%d, %d]" % x_window(bcpos, q))
    o.append("\"rows\": [\n%s],\n" % ",\n".join(rows))
    o.append("\"xranges\": [\n%s]\n" % ",\n".join(xr))
    o.append("}\n}\n")
    return "".join(o)


def sort_variants(var):
    var.sort(key=lambda v: (v["chr"], v["pos"], v["basenum"]))
