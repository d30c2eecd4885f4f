"""`tracy assemble` (assemble.h:56-589) restated in Python over the oracle -- tests only; cross-checks the `assemble`
command of tracy_amd/cli.  PARITY UNPINNED (assemble.h needs Boost / sdsl)."""
import os

import numpy as np

import msa_oracle as mo
import pyoracle as orc
import sage_oracle as so

_RC = dict(zip("ACGTNHVMYDBKRUSW", "TGCANDBKRHVMYASW"))


def load(path, pratio, stringency):
    from tracy_amd import hostlib
    t = hostlib.read_trace(path)
    sig, pos = t["signal"], t["basecallpos"]
    pri, sec, con, bcpos, q = hostlib.basecall_qual(sig, pos, pratio)
    tl = tr = 0
    if stringency:
        tl, tr = so.trim_trace(stringency, sec, bcpos.tolist())
    prof = orc.create_profile_trace(sig, bcpos, pri, sec, tl, tr)
    return dict(sig=sig, bcpos=bcpos.tolist(), pri=pri, sec=sec, con=con, q=q.tolist(), tl=tl, tr=tr, prof=prof,
                stem=os.path.splitext(os.path.basename(path))[0])


def hard_trim(t):
    """trim.h:75-98 -> basecall dict"""
    n = len(t["pri"])
    ln = n - t["tr"]
    out = dict(bcPos=[], primary=bytearray(), secondary=bytearray(), consensus=bytearray(), estQual=[])
    bc, idx = 0, t["bcpos"][0]
    for x in range(t["sig"].shape[1]):
        if idx == x:
            if t["tl"] <= bc < ln:
                out["bcPos"].append(x)
                out["primary"].append(t["pri"][bc])
                out["secondary"].append(t["sec"][bc])
                out["consensus"].append(t["con"][bc])
                out["estQual"].append(t["q"][bc])
            if bc < n - 1:
                bc += 1
                idx = t["bcpos"][bc]
    return out


def revcomp_trace(sig, nbc):
    """trim.h:123-149"""
    ns = sig.shape[1]
    out = dict(bcPos=[], primary=bytearray(), secondary=bytearray(), consensus=bytearray(), estQual=[])
    bc = len(nbc["bcPos"]) - 1
    idx = nbc["bcPos"][bc]
    newpos = 0
    for x in range(ns, 0, -1):
        if idx == x - 1:
            out["bcPos"].append(newpos)
            for f in ("primary", "secondary", "consensus"):
                ch = chr(nbc[f][bc])
                out[f].append(ord(_RC.get(ch, ch)))
            out["estQual"].append(nbc["estQual"][bc])
            if bc > 0:
                bc -= 1
                idx = nbc["bcPos"][bc]
        newpos += 1
    nsig = np.ascontiguousarray(sig[::-1, ::-1])
    return nsig, out


def gapped_trace(t, forward, row, name):
    nbc = hard_trim(t)
    sig = t["sig"]
    if not forward:
        sig, nbc = revcomp_trace(sig, nbc)
    padded = so.alignment_trace_padding(row.encode(), sig, nbc["bcPos"], bytes(nbc["primary"]), bytes(nbc["secondary"]),
                                        bytes(nbc["consensus"]), nbc["estQual"])
    return so.assembly_trace_text(padded, name)


def aligned_trace_by_row(row, name, forward, ref):
    """json.h:220-246"""
    lead = len(row) - len(row.lstrip("-"))
    trail = len(row) - len(row.rstrip("-"))
    body = row[lead:len(row) - trail] if lead < len(row) else ""
    return ("{\n\"reference\": %s,\n\"forward\": %s,\n\"traceFileName\": \"%s\",\n\"leadingGaps\": \"%d\",\n\"trailingGaps\": \"%d\",\n"
            "\"align\": \"%s\"\n}\n" % ("true" if ref else "false", "true" if forward else "false", name, lead, trail, body))


def rows_of(p1, p2, btr):
    ops = btr[::-1].decode()
    r0, r1, x, y = [], [], 0, 0
    for op in ops:
        if op != "h":
            r0.append(mo.cons_char(p1, x))
            x += 1
        else:
            r0.append("-")
        if op != "v":
            r1.append(mo.cons_char(p2, y))
            y += 1
        else:
            r1.append("-")
    return "".join(r0), "".join(r1), ops


def common_files(align, gapped, cs, qstr, fmt):
    files = {".vertical": "".join("".join(r[j] for r in align) + "|" + gapped[j] + "\n" for j in range(len(align[0]) if align else 0))}
    if fmt == "fasta":
        files[".cons.fa"] = ">Consensus\n%s\n" % cs
    else:
        files[".cons.fq"] = "@Consensus\n%s\n+\n%s\n" % (cs, qstr)
    return files


def assemble_ref_guided(paths, ref_path, score, pratio=0.33, stringency=4, fracmatch=0.5, called=0.1, inccons=False, incref=False, fmt="fasta"):
    name, seq = so.load_single_fasta(ref_path)
    pref = orc.create_profile_str(seq.encode())
    traces = [load(p, pratio, stringency) for p in paths]
    profiles, score_idx = [], []
    f32 = np.float32
    for i, t in enumerate(traces):
        rev = orc.revcomp_profile(t["prof"])
        gf = orc.gotoh_score_prof(t["prof"], pref, 1, 0, score)
        gr = orc.gotoh_score_prof(rev, pref, 1, 0, score)
        size = float(t["prof"].shape[1])
        thr = size * float(f32(fracmatch)) * score[0] + size * float(f32(1) - f32(fracmatch)) * score[1]
        if gf > thr or gr > thr:
            fwd = gf >= gr
            score_idx.append(dict(score=max(gf, gr), idx=i, newidx=len(score_idx), forward=fwd))
            profiles.append(t["prof"] if fwd else rev)
    score_idx.sort(key=lambda s: (-s["score"], s["idx"]))
    if not score_idx:
        return common_files([], "", "", "", fmt), []
    p0 = profiles[score_idx[0]["newidx"]]
    _, btr = orc.gotoh_prof(p0, pref, 1, 0, score)
    r0, r1, _ = rows_of(p0, pref, btr)
    align = [r0, r1]
    for s in score_idx[1:]:
        ap = mo.profile_of_alignment(align)
        pn = profiles[s["newidx"]]
        _, btr = orc.gotoh_prof(pn, np.ascontiguousarray(ap), 1, 0, score)
        new0, _, ops = rows_of(pn, ap, btr)
        comb = [list(new0)] + [[] for _ in align]
        a = 0
        for j, op in enumerate(ops):
            if op != "v":
                for k in range(len(align)):
                    comb[k + 1].append(align[k][a])
                a += 1
            else:
                for k in range(len(align)):
                    comb[k + 1].append("-")
        align = ["".join(r) for r in comb]
    gapped, cs, qstr = mo.consensus(align, called, not incref)
    n = len(score_idx)
    fa = []
    for i, s in enumerate(score_idx):
        fa.append(">%s %s\n%s\n" % (traces[s["idx"]]["stem"], "(forward)" if s["forward"] else "(reverse)", align[n - i - 1]))
    fa.append(">Reference\n%s\n" % align[n])
    if inccons:
        fa.append(">Consensus\n%s\n" % gapped)
    js = ["{\n\"gapFreeConsensus\": \"%s\",\n\"gappedConsensus\": \"%s\",\n\"msa\": \n[\n" % (cs, gapped)]
    js.append(",\n".join(aligned_trace_by_row(align[n - i - 1], traces[s["idx"]]["stem"], s["forward"], False) for i, s in enumerate(score_idx)))
    js.append(",\n" + aligned_trace_by_row(align[n], "", True, True))
    js.append("],\n\"gappedTraces\": \n[\n")
    js.append(", ".join(gapped_trace(traces[s["idx"]], s["forward"], align[n - i - 1], traces[s["idx"]]["stem"]) for i, s in enumerate(score_idx)))
    js.append("]\n}\n")
    files = common_files(align, gapped, cs, qstr, fmt)
    files[".align.fa"] = "".join(fa)
    files[".json"] = "".join(js)
    return files, score_idx


def assemble_denovo(paths, score, pratio=0.33, stringency=4, fracmatch=0.5, called=0.1, inccons=False, fmt="fasta"):
    traces = [load(p, pratio, stringency) for p in paths]
    profs, fwdp = mo.rev_seq_based_on_dist([t["prof"] for t in traces], score)
    f32 = np.float32
    keep = []
    for i in range(len(profs)):
        hit = False
        for j in range(len(profs)):
            if i == j:
                continue
            gs, btr = orc.gotoh_prof(profs[i], profs[j], 1, 1, score)
            na = btr.count(b"s")
            frac = na / float(profs[i].shape[1])
            thr = float(f32(f32(f32(na) * f32(fracmatch)) * f32(score[0])) + f32(f32(f32(na) * (f32(1) - f32(fracmatch))) * f32(score[1])))
            if frac > 0.1 and na > 25 and gs > thr:
                hit = True
                break
        if hit:
            keep.append(i)
    if len(keep) < 2:
        return None, None
    sps = [profs[i] for i in keep]
    fwd = [fwdp[i] for i in keep]
    align, seqidx = mo.msa(sps, score)
    gapped, cs, qstr = mo.consensus(align, called, False)
    name = lambda r: traces[keep[seqidx[r]]]["stem"]
    fa = "".join(">%s %s\n%s\n" % (name(r), "(forward)" if fwd[seqidx[r]] else "(reverse)", align[r]) for r in range(len(align)))
    if inccons:
        fa += ">Consensus\n%s\n" % gapped
    js = ["{\n\"gapFreeConsensus\": \"%s\",\n\"gappedConsensus\": \"%s\",\n\"msa\": \n[\n" % (cs, gapped)]
    js.append(",\n".join(aligned_trace_by_row(align[r], name(r), fwd[seqidx[r]], False) for r in range(len(align))))
    js.append("],\n\"gappedTraces\": \n[\n")
    js.append(", ".join(gapped_trace(traces[keep[seqidx[r]]], fwd[seqidx[r]], align[r], name(r)) for r in range(len(align))))
    js.append("]\n}\n")
    files = common_files(align, gapped, cs, qstr, fmt)
    files[".align.fa"] = fa
    files[".json"] = "".join(js)
    return files, dict(keep=keep, fwd=fwd, seqidx=seqidx)
