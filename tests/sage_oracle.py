"""The `tracy align` hot section (sage.h:191-311) composed from oracle functions -- the expected
behaviour of tracyhip_align_traces (tests only)."""
import numpy as np

import pyoracle as orc

_COMP = {ord("A"): "T", ord("C"): "G", ord("G"): "C", ord("T"): "A", ord("N"): "N"}


def revcomp(s):
    """reverseComplement(std::string), fmindex.h:8-24, for [ACGTN] input"""
    return "".join(_COMP[c] for c in reversed(s)).encode()


def align_trace(profile_full, ref, score, trim_left=50, trim_right=50):
    mf = profile_full.shape[1]
    tl, tr = trim_left, trim_right
    if tl + tr >= mf:
        tl = tr = 0
    trimmed = np.ascontiguousarray(profile_full[:, tl:mf - tr])
    fwdp = orc.create_profile_str(ref)
    revp = orc.revcomp_profile(fwdp)
    gs_fwd = orc.gotoh_score_prof(trimmed, fwdp, 1, 0, score)
    gs_rev = orc.gotoh_score_prof(trimmed, revp, 1, 0, score)
    forward = gs_fwd > gs_rev
    refslice = ref if forward else revcomp(ref)
    pref = fwdp if forward else revp
    sc1, btr1 = orc.gotoh_prof(trimmed, pref, 1, 0, score)
    r0, r1 = orc.create_alignment_prof(btr1, trimmed, pref)
    ri, risize, pos_add, _ = orc.trim_reference_slice(r0, r1, trim_left, trim_right, len(refslice), forward)
    sl = refslice[ri:ri + risize]
    refprof = orc.create_profile_str(sl)
    sc2, btr2 = orc.gotoh_prof(profile_full, refprof, 1, 0, score)
    return dict(score_fwd=gs_fwd, score_rev=gs_rev, forward=int(forward), score_prelim=sc1, slice_begin=ri,
                slice_len=len(sl), ref_pos=pos_add, score_final=sc2, btr=btr2, refslice=sl)
