"""The `tracy decompose` hot section (indigo.h:190-388, FASTA reference) composed from oracle functions --
the expected behaviour of tracyhip_decompose_traces (tests only)."""
import numpy as np

import pyoracle as orc
from sage_oracle import revcomp


def trimmed_seq(s, ltrim, rtrim):
    if ltrim + rtrim + 1 >= len(s):
        return s
    return s[ltrim:len(s) - rtrim]


def decompose_trace(sig, bcpos, pri, sec, ref, score, tl=50, tr=50, maxindel=1000, madc=5):
    trimmed = orc.create_profile_trace(sig, bcpos, pri, sec, tl, tr)
    bp = orc.find_breakpoint(trimmed)
    fwdp = orc.create_profile_str(ref)
    revp = orc.revcomp_profile(fwdp)
    gs_fwd = orc.gotoh_score_prof(trimmed, fwdp, 1, 0, score)
    gs_rev = orc.gotoh_score_prof(trimmed, revp, 1, 0, score)
    forward = gs_fwd > gs_rev
    refslice = ref if forward else revcomp(ref)
    pref = fwdp if forward else revp
    sc1, btr1 = orc.gotoh_prof(trimmed, pref, 1, 0, score)
    rows = orc.create_alignment_prof(btr1, trimmed, pref)
    seqsize = float(trimmed.shape[1])
    status = 0
    if sc1 <= seqsize * 0.35 * score[0] + seqsize * (1 - 0.35) * score[1]:
        status = -1
    if not bp.indelshift:
        rc, bp = orc.find_homozygous_breakpoint(rows[0], rows[1], bp)
        if rc != 1 and status == 0:
            status = -2
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
