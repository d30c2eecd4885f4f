"""Synthetic `tracy decompose` cases shared by the CPU (emulator) and GPU parity tests: heterozygous
indels, SNV-only traces and hand-made alignments, each carried through the oracle's indigo.h chain."""
import numpy as np

import pyoracle as orc

SC = (3, -5, -10, -4)


def make_case(seed, n=1600, mf=520, kind=0, frac1=0.6, maxlen=30):
    from tracy_amd import hostlib
    ref, sig, pos, indel = hostlib.synth_decompose(seed, n, mf, maxlen, kind, frac1)
    pri, sec, con, bcpos = hostlib.basecall(sig, pos, 0.33)
    prof = hostlib.create_profile(sig, bcpos, pri, sec, 50, 50)
    bp = orc.find_breakpoint(prof)
    fwd = orc.create_profile_str(ref)
    score, btr = orc.gotoh_prof(prof, fwd, 1, 0, SC)
    r0, r1 = orc.create_alignment_prof(btr, prof, fwd)
    hom_rc = 1
    if not bp.indelshift:
        hom_rc, bp = orc.find_homozygous_breakpoint(r0, r1, bp)
    return dict(ref=ref, sig=sig, pos=pos, indel=indel, pri=pri, sec=sec, con=con, bcpos=bcpos, prof=prof, bp=bp,
                rows=(r0, r1), score=score, hom_rc=hom_rc)


def oracle_decompose(c, maxindel=1000, madc=5):
    p2, s2, dcp, st = orc.decompose_alleles(c["rows"][0], c["rows"][1], c["pri"], c["sec"], c["bp"], len(c["ref"]),
                                            50, 50, maxindel, madc)
    sd = orc.generate_secondary_decomposed(c["sig"], c["bcpos"], p2, s2)
    af = orc.allelic_fraction(c["sig"], c["bcpos"], p2, sd, 50, 50)
    return dict(pri=p2, sec=s2, dcp=dcp, status=st, secdecomp=sd, af=af)


def case_list():
    cases = []
    for seed, kind, frac in [(7, 0, 0.6), (8, 0, 0.55), (9, 0, 0.7), (10, 1, 0.6), (11, 0, 0.5), (12, 1, 0.5), (13, 0, 0.65),
                             (14, 0, 0.6)]:
        cases.append(make_case(seed, kind=kind, frac1=frac, maxlen=(30 if seed != 14 else 3)))
    return cases
