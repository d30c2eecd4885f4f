"""`tracy consensus` (consensus.h) restated in Python over the oracle -- tests only.  PARITY UNPINNED."""
import math

import numpy as np

import assemble_oracle as ao
import pyoracle as orc
import sage_oracle as so


def _round(x):
    """boost::math::round / std::round: half away from zero"""
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def gt_letter(cl, use_iupac):
    """consensus.h:94-171 -> (letter, quality)"""
    total = 0.0
    for v in cl:
        total += v
    gl = []
    for k in range(6):
        c = cl[k] / total if total > 0 else 0.0
        g = max(math.log10(c), -1000.0) if c > 0 else -1000.0
        gl.append(g)
    best, second = (1, 0) if gl[0] < gl[1] else (0, 1)
    for k in range(2, 6):
        if gl[k] > gl[best]:
            second, best = best, k
        elif gl[k] > gl[second]:
            second = k
    ambiguous = use_iupac and gl[second] > -1 and best <= 3 and second <= 3
    bv = gl[best]
    gl = [g - bv for g in gl]
    bpl = int(_round(-10 * gl[best])) & 0xFFFFFFFF
    spl = int(_round(-10 * gl[second])) & 0xFFFFFFFF
    x = 1 - 1 / (math.pow(10.0, -(bpl / 10.0)) + math.pow(10.0, -(spl / 10.0)))
    like = math.log10(x) if x > 0 else float("-inf")  # C's log10(0) is -inf
    like = like if like > -1000 else -1000.0
    gq = max(int(_round(-10 * like)), 0)
    if ambiguous:
        import ctypes
        letter = orc.lib().orc_iupac2(ctypes.c_char("ACGT"[best].encode()), ctypes.c_char("ACGT"[second].encode())).decode()
    else:
        letter = "ACGTN-"[best]
    return letter, gq


def pairwise_consensus(row0, row1, p1, p2, union=True, iupac=False):
    cons, qual, s1, s2 = [], [], 0, 0
    for a, b in zip(row0, row1):
        if a == "-" or b == "-":
            if a != "-":
                if union:
                    l, q = gt_letter([float(p1[k, s1]) for k in range(6)], iupac)
                    cons.append(l); qual.append(q)
                s1 += 1
            if b != "-":
                if union:
                    l, q = gt_letter([float(p2[k, s2]) for k in range(6)], iupac)
                    cons.append(l); qual.append(q)
                s2 += 1
        else:
            l, q = gt_letter([float(np.float32(p1[k, s1]) + np.float32(p2[k, s2])) for k in range(6)], iupac)
            cons.append(l); qual.append(q)
            s1 += 1
            s2 += 1
    return "".join(cons), qual


def plot_clustal_pairwise(row0, row1, stem1, stem2, forward, score, linelimit=60):
    fald = linelimit + 14
    o = []

    def seq_block(row):
        count = 0
        for ch in row:
            if ch != "-":
                o.append(ch)
                if (count + 1) % fald == 0:
                    o.append("\n")
                count += 1
        if count % fald != 0:
            o.append("\n")
    o.append(">%s\n" % stem1)
    seq_block(row0)
    o.append(">%s %s\n" % (stem2, "(forward)" if forward else "(reverse)"))
    seq_block(row1)
    o.append("\nAlignment score: %d\n" % score)
    o.append("#" + "-" * (fald - 1) + "\n\n")
    f1, f2 = stem1[:8].ljust(8), stem2[:8].ljust(8)
    vi = ri = 1
    blocks, s = 0, 0
    while s < len(row0):
        a, b = row0[s:s + linelimit], row1[s:s + linelimit]
        o.append("%s%5d %s\n" % (f1, vi, a))
        vi += sum(1 for ch in a if ch != "-")
        o.append(" " * 14 + "".join("|" if x == y else " " for x, y in zip(a, b)) + "\n")
        o.append("%s%5d %s\n\n" % (f2, ri, b))
        ri += sum(1 for ch in b if ch != "-")
        s += linelimit
        blocks += 1
    for _ in range(blocks, 6):
        o.append("\n" * 4)
    o.append(("#" + "-" * (fald - 1) + "\n") * 2)
    o.append("\n\n")
    return "".join(o)


def consensus(path1, path2, score, trims=(50, 50, 50, 50), pratio=0.33, label="Consensus", union=True, iupac=False, linelimit=60,
              min_overlap=25, fracmatch=0.5):
    from tracy_amd import hostlib
    t = []
    for p in (path1, path2):
        r = hostlib.read_trace(p)
        pri, sec, con, bcpos, q = hostlib.basecall_qual(r["signal"], r["basecallpos"], pratio)
        t.append(dict(sig=r["signal"], pos=r["basecallpos"], pri=pri, sec=sec, bcpos=bcpos))
    p1 = orc.create_profile_trace(t[0]["sig"], t[0]["bcpos"], t[0]["pri"], t[0]["sec"], trims[0], trims[1])
    f2 = orc.create_profile_trace(t[1]["sig"], t[1]["bcpos"], t[1]["pri"], t[1]["sec"], trims[2], trims[3])
    r2 = orc.revcomp_profile(f2)
    forward = orc.gotoh_score_prof(p1, f2, 1, 1, score) > orc.gotoh_score_prof(p1, r2, 1, 1, score)
    p2 = f2 if forward else r2
    sc, btr = orc.gotoh_prof(p1, np.ascontiguousarray(p2), 1, 1, score)
    row0, row1, _ = ao.rows_of(p1, p2, btr)
    aligned = sum(1 for a, b in zip(row0, row1) if a != "-" and b != "-")
    matches = sum(1 for a, b in zip(row0, row1) if a != "-" and b != "-" and a == b)
    if aligned < min_overlap or (matches / aligned if aligned else 0.0) < float(np.float32(fracmatch)):
        return None
    import os
    s1, s2 = (os.path.splitext(os.path.basename(p))[0] for p in (path1, path2))
    cons, qual = pairwise_consensus(row0, row1, p1, p2, union, iupac)
    files = {
        ".align.fa": ">%s\n%s\n>%s %s\n%s\n" % (s1, row0, s2, "(forward)" if forward else "(reverse)", row1),
        ".fa": ">%s\n%s\n" % (label, cons),
        ".fq": "@%s\n%s\n+\n%s\n" % (label, cons, "".join(chr(min(q + 33, 122)) for q in qual)),
        ".txt": plot_clustal_pairwise(row0, row1, s1, s2, forward, sc, linelimit),
    }
    return files, forward
