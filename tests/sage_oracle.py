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


# ---------------------------------------------------------------------------------------------------------
# `tracy align` host stages around the DP, restated in Python straight from the reference text (tests only;
# independent of tracy_amd/host/sage_out.hpp, which it cross-checks).  PARITY UNPINNED: the reference
# headers need Boost/htslib/sdsl and cannot be compiled here.
# ---------------------------------------------------------------------------------------------------------
def _u32(x):
    return x & 0xFFFFFFFF


def _i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def is_ambiguous(c):
    return c not in b"ACGT"


def find_best_trace_section(secondary, bcpos, win=10):
    """abif.h:164-220 -> (penalty list, best index, per-base penalty)"""
    n = len(secondary)
    pen = [0] * n
    half = win // 2
    amb = sum(1 for i in range(min(win, n)) if is_ambiguous(secondary[i:i + 1]))
    for i in range(min(half, n)):
        pen[i] = amb
    for i in range(win, n):
        if is_ambiguous(secondary[i - win:i - win + 1]):
            amb -= 1
        if is_ambiguous(secondary[i:i + 1]):
            amb += 1
        pen[i - half] = amb
    if n >= half:
        for i in range(n - half, n):
            pen[i] = amb
    mean = 0.0
    for i in range(1, n):
        mean += bcpos[i] - bcpos[i - 1]
    denom = float(_u32(n - 1)) if n != 0 else float(2 ** 64 - 1)
    mean = mean / denom if denom != 0 else (float("nan") if mean == 0 else float("inf"))
    peak_var = 0
    i = 0
    while i + win < n:
        old = bcpos[i - 1] if i > 0 else 0
        lo, hi = _u32(bcpos[n - 1]), 0
        for k in range(win):
            d = _u32(bcpos[i + k] - old)
            old = bcpos[i + k]
            lo, hi = min(lo, d), max(hi, d)
        peak_var = _u32(int((abs(hi - mean) + abs(lo - mean)) / 2))
        pen[i + half] = _i32(pen[i + half] + peak_var)
        if i == 0:
            for k in range(half):
                pen[k] = _i32(pen[k] + peak_var)
        i += 1
    if n >= half:
        for i in range(n - half, n):
            pen[i] = _i32(pen[i] + peak_var)
    source = int(0.1 * n)
    best_idx, best_val = 0, 99999999
    i = 0
    while i + source < n:
        v = sum(pen[i:i + source])
        if v < best_val:
            best_val, best_idx = v, i + source // 2
        i += 1
    per_base = best_val / source if source else float("inf")
    return pen, best_idx, per_base


def trim_trace(stringency, secondary, bcpos):
    """trim.h:35-73 -> (leftTrim, rightTrim)"""
    import numpy as np
    win = 10
    n = len(secondary)
    pen, best, per_base = find_best_trace_section(secondary, bcpos, win)
    limit = float(np.float32(stringency)) * per_base
    right, left = n, 0
    local = float(sum(pen[best:min(best + win, n)]))
    i = best
    while i + win < n:
        local -= pen[i]
        local += pen[i + win]
        if local > limit * win:
            right = i
            break
        i += 1
    local = float(sum(pen[best:min(best + win, n)]))
    i = best - 1
    while i >= 0:
        if i + win < n:
            local -= pen[i + win]
        local += pen[i]
        if local > limit * win:
            left = i + win - 1
            break
        i -= 1
    right = n - right if right < n else 0
    return left, right


def load_single_fasta(path):
    """fasta.h:54-95 -> (name, seq) or None"""
    name, body = "", ""
    for line in open(path, "rb").read().decode("latin1").split("\n"):
        if not line:
            continue
        if line[0] == ">":
            if name:
                return None
            name = line[1:-1] if line.endswith("\r") else line[1:]
        else:
            body += (line[:-1] if line.endswith("\r") else line).upper()
    out = []
    for ch in body:
        if ch in "ACGTN":
            out.append(ch)
        elif ch in "WSMKRYBDHV":
            out.append("N")
        else:
            return None
    for bad in "\\,'\"()[]{}<>:\t\r#":
        name = name.replace(bad, "")
    return name, "".join(out)


def _fmt_double(x):
    return "%g" % x


def plot_alignment(row0, row1, chrom, pos, refslice_len, forward, score, linelimit=60, key=0, a1a2=(0.0, 0.0)):
    """fmindex.h:329-427 -> file text"""
    r0, r1 = row0.decode(), row1.decode()
    o = []
    ri, riend, vi = pos + 1, pos + refslice_len, 1
    fald = linelimit + 14

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
    if key == 0:
        o.append(">Alt\n")
    elif key == 2:
        o.append(">Alt2 (Estimated allelic Fraction: %s)\n" % _fmt_double(a1a2[1]))
    else:
        o.append(">Alt1 (Estimated allelic Fraction: %s)\n" % _fmt_double(a1a2[0]))
    seq_block(r0)
    if key != 3:
        if forward:
            o.append(">Ref %s:%d-%d forward\n" % (chrom, ri, riend))
        else:
            o.append(">Ref %s:%d-%d reversecomplement\n" % (chrom, pos + refslice_len - (riend - pos) + 1, pos + refslice_len - (ri - pos) + 1))
    else:
        o.append(">Alt2 (Estimated allelic Fraction: %s)\n" % _fmt_double(a1a2[1]))
    seq_block(r1)
    o.append("\n")
    o.append("Alignment score: %d\n" % score)
    o.append("#" + "-" * (fald - 1) + "\n\n")
    blocks, s, e = 0, 0, len(r0)
    while s < e:
        seg0, seg1 = r0[s:s + linelimit], r1[s:s + linelimit]
        o.append(("Alt%10d " % vi) if key != 3 else ("Alt1%9d " % vi))
        o.append(seg0 + "\n")
        vi += sum(1 for ch in seg0 if ch != "-")
        o.append(" " * 14 + "".join("|" if a == b else " " for a, b in zip(seg0, seg1)) + "\n")
        if key != 3:
            o.append("Ref%10d " % (ri if forward else pos + refslice_len - (ri - pos) + 1))
        else:
            o.append("Alt2%9d " % ri)
        o.append(seg1 + "\n\n")
        ri += sum(1 for ch in seg1 if ch != "-")
        s += linelimit
        blocks += 1
    for _ in range(blocks, 6):
        o.append("\n" * 4)
    o.append(("#" + "-" * (fald - 1) + "\n") * 2)
    o.append("\n\n")
    return "".join(o)


def alignment_trace_padding(row, signal, bcpos, primary, secondary, consensus, estqual):
    """json.h:383-472 -> dict(signal [4][ns'], bcPos, primary, secondary, consensus, estQual, leadingGaps, trailingGaps)"""
    step = 6
    if len(bcpos) > 1:
        avg = 0.0
        for i in range(1, len(bcpos)):
            avg += bcpos[i] - bcpos[i - 1]
        step = int(avg / (len(bcpos) - 1))
    ins_pos, ins_size = [], []
    pos, ingap, gapsize, leading = 0, False, 0, 0
    for ch in row:
        if ch == ord("-"):
            gapsize = gapsize + 1 if ingap else 1
            ingap = True
        else:
            if ingap:
                ingap = False
                if pos:
                    ins_pos.append(int((bcpos[pos - 1] + bcpos[pos]) / 2.0))
                    ins_size.append(gapsize)
                else:
                    leading = gapsize
            pos += 1
    trailing = gapsize if ingap else 0
    out = [[], [], [], []]
    nb = dict(bcPos=[], primary=bytearray(), secondary=bytearray(), consensus=bytearray(), estQual=[])
    bc, idx, offset, ins = 0, bcpos[0], 0, 0
    ins_idx = ins_pos[0] if ins_pos else -1
    for x in range(len(signal[0])):
        for k in range(4):
            out[k].append(int(signal[k][x]))
        if ins_idx == x:
            for _ in range(ins_size[ins]):
                nb["bcPos"].append(x + offset + int(step / 2.0))
                nb["estQual"].append(0)
                for f in ("primary", "secondary", "consensus"):
                    nb[f].append(ord("-"))
                for _s in range(step):
                    for k in range(4):
                        out[k].append(-99)
                    offset += 1
            if ins < len(ins_pos) - 1:
                ins += 1
                ins_idx = ins_pos[ins]
        if idx == x:
            nb["bcPos"].append(idx + offset)
            nb["estQual"].append(int(estqual[bc]))
            nb["primary"].append(primary[bc])
            nb["secondary"].append(secondary[bc])
            nb["consensus"].append(consensus[bc])
            if bc < len(bcpos) - 1:
                bc += 1
                idx = bcpos[bc]
    nb.update(signal=out, leadingGaps=leading, trailingGaps=trailing)
    return nb


def assembly_trace_text(padded, name="trace"):
    """assemblyTrace, json.h:108-194 -> text of one gapped-trace object"""
    sig, bcpos = padded["signal"], padded["bcPos"]
    ns = len(sig[0])
    o = ["{\n", "\"traceFileName\": \"%s\",\n" % name, "\"leadingGaps\": %d,\n" % padded["leadingGaps"],
         "\"trailingGaps\": %d,\n" % padded["trailingGaps"]]
    for k, nm in enumerate(("peakA", "peakC", "peakG", "peakT")):
        o.append("\"%s\": [%s],\n" % (nm, ", ".join(str(v) for v in sig[k])))

    def visited():
        """(sample index, call index) pairs the reference's cursor loop emits"""
        bc, idx, res = 0, bcpos[0], []
        for i in range(ns):
            if idx == i:
                res.append((i, bc))
                if bc < len(bcpos) - 1:
                    bc += 1
                    idx = bcpos[bc]
        return res
    vis = visited()

    def joined(items):
        # the separator is written before every item whose sample index differs from bcPos[0]
        return "".join((", " if i != bcpos[0] else "") + txt for (i, txt) in items)
    o.append("\"basecallPos\": [%s],\n" % joined([(i, str(i + 1)) for i, _ in vis]))
    o.append("\"basecallQual\": [%s],\n" % joined([(i, str(padded["estQual"][b])) for i, b in vis]))
    items, gapless = [], 0
    for i, b in vis:
        p, s = chr(padded["primary"][b]), chr(padded["secondary"][b])
        if p != "-":
            gapless += 1
            items.append((i, "\"%d\":\"%d:%s%s\"" % (i + 1, gapless, p, ("|" + s) if p != s else "")))
        else:
            items.append((i, "\"%d\":\"-\"" % (i + 1)))
    o.append("\"basecalls\": {%s}\n" % joined(items))
    o.append("}\n")
    return "".join(o)


def trace_align_json(padded, chrom, pos, forward, row0, row1):
    """json.h:197-217 -> file text"""
    o = ["{\n", "\"gappedTrace\":\n", assembly_trace_text(padded, "trace")]
    o.append(",\n")
    o.append("\"refchr\": \"%s\",\n" % chrom)
    o.append("\"refpos\": %d,\n" % (pos + 1))
    o.append("\"altalign\": \"%s\",\n" % row0.decode())
    o.append("\"refalign\": \"%s\",\n" % row1.decode())
    o.append("\"forward\": %d\n" % (1 if forward else 0))
    o.append("}\n")
    return "".join(o)


def align_fasta_text(stem, chrom, forward, row0, row1):
    """sage.h:328-339"""
    return ">%s\n%s\n>%s %s\n%s\n" % (stem, row0.decode(), chrom, "(forward)" if forward else "(reverse)", row1.decode())


# ---------------------------------------------------------------------------------------------------------
# k-mer seeding in an indexed genome (fmindex.h:173-326), brute-force restatement over the "dump" text
# (upper-cased contigs joined by newlines).  Tests only; cross-checks tracy_amd/host/seed.hpp.
# ---------------------------------------------------------------------------------------------------------
class BruteGenome:
    def __init__(self, contigs):
        """contigs: list of (name, sequence str)"""
        self.names = [n for n, _ in contigs]
        self.lengths = [len(s) for _, s in contigs]
        self.text = "".join(s.upper() + "\n" for _, s in contigs)
        self._cache = {}

    def locate(self, pat):
        if pat not in self._cache:
            out, p = [], self.text.find(pat)
            while p != -1:
                out.append(p)
                p = self.text.find(pat, p + 1)
            self._cache[pat] = out
        return self._cache[pat]


def _revcomp_str(s):
    """reverseComplement(std::string&), fmindex.h:11-25: letters outside ACGTN keep the ORIGINAL byte of that position"""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    rev = s[::-1].upper()
    return "".join(comp.get(rev[i], s[i]) for i in range(len(s)))


def scan_sequence(g, consensus, trim_left, trim_right, kmer, unique):
    hits = []
    ncount = sum(1 for i in range(trim_left, min(trim_left + kmer, len(consensus))) if consensus[i] == "N")
    k = trim_left
    bound = (len(consensus) - trim_right) % (1 << 64)  # fmindex.h:211: size() - trimRight in size_t wraps when the trim is the longer one
    while k < bound and k < len(consensus):
        if ncount == 0:
            seq = consensus[k:k + kmer]
            loc = g.locate(seq)
            if unique:
                if len(loc) == 1:
                    hits.append(loc[0] - k)
            elif 0 < len(loc) < 1000:
                hits.extend(p - k for p in loc)
        if consensus[k] == "N":
            ncount -= 1
        if k + kmer < len(consensus) and consensus[k + kmer] == "N":
            ncount += 1
        k += 1
    return hits


def find_max_freq(hits):
    if not hits:
        return 0, 0
    hits = sorted(hits)
    best, gpos, run = 1, hits[0], 1
    for i in range(1, len(hits)):
        if hits[i] == hits[i - 1]:
            run += 1
            if run > best:
                best, gpos = run, hits[i]
        else:
            run = 1
    return best, gpos


def get_reference_slice(g, consensus, trim_left=50, trim_right=50, kmer=15, min_support=3, maxindel=1000):
    """fmindex.h:236-326 for an indexed genome -> dict(forward, kmersupport, pos, chr, refslice) or None"""
    rv = _revcomp_str(consensus)
    res = None
    for unique in (True, False):
        ff, bf = find_max_freq(scan_sequence(g, consensus, trim_left, trim_right, kmer, unique))
        fr, br = find_max_freq(scan_sequence(g, rv, trim_right, trim_left, kmer, unique))
        if ff >= min_support and ff > 2 * fr:
            res = (True, ff, bf)
            break
        if fr >= min_support and fr > 2 * ff:
            res = (False, fr, br)
            break
    if res is None:
        return None
    forward, support, best = res
    cumsum, ref = 0, 0
    while best >= cumsum + g.lengths[ref] + 1:
        cumsum += g.lengths[ref] + 1
        ref += 1
    seqlen = g.lengths[ref] + 1
    chrpos = max(best - cumsum, 0)
    slicestart = chrpos - maxindel if chrpos > maxindel else 0
    sliceend = seqlen
    tmpend = chrpos + len(consensus) + maxindel
    if tmpend < seqlen:
        sliceend = tmpend
    last = min(sliceend, g.lengths[ref] - 1)  # faidx_fetch_seq: inclusive end, clipped
    start = sum(n + 1 for n in g.lengths[:ref])
    sl = g.text[start + slicestart:start + last + 1] if slicestart <= last else ""
    if not forward:
        sl = _revcomp_str(sl)
    return dict(forward=forward, kmersupport=support, pos=slicestart, chr=g.names[ref], contig=ref, refslice=sl)


def align_trace_oriented(profile_full, window, forward, score, trim_left=50, trim_right=50):
    """the indexed-genome branch of sage.h (:217-221, 258-260, 311): `window` is already oriented"""
    import numpy as np
    mf = profile_full.shape[1]
    tl, tr = trim_left, trim_right
    if tl + tr >= mf:
        tl = tr = 0
    trimmed = np.ascontiguousarray(profile_full[:, tl:mf - tr])
    pref = orc.create_profile_str(window)
    sc1, btr1 = orc.gotoh_prof(trimmed, pref, 1, 0, score)
    r0, r1 = orc.create_alignment_prof(btr1, trimmed, pref)
    ri, risize, pos_add, _ = orc.trim_reference_slice(r0, r1, trim_left, trim_right, len(window), forward)
    sl = window[ri:ri + risize]
    sc2, btr2 = orc.gotoh_prof(profile_full, orc.create_profile_str(sl), 1, 0, score)
    return dict(score_prelim=sc1, slice_begin=ri, slice_len=len(sl), ref_pos=pos_add, score_final=sc2, btr=btr2, refslice=sl)
