"""`tracy assemble`'s multiple alignment (msa.h) restated in Python over the oracle's Gotoh functions -- tests only;
cross-checks tracy_amd/host/msa.hpp.  PARITY UNPINNED (msa.h needs Boost)."""
import numpy as np

import pyoracle as orc


def cons_char(p, j):
    k = 0
    best = float(p[0, j])
    for r in range(1, 6):
        if float(p[r, j]) > best:
            best, k = float(p[r, j]), r
    return "ACGTNN"[k]


def profile_of_alignment(rows):
    """align.h:138-180"""
    nrow, ncol = len(rows), len(rows[0])
    p = np.zeros((6, ncol), np.float32)
    first = [-1] * nrow
    last = [ncol] * nrow
    for i, r in enumerate(rows):
        for j, ch in enumerate(r):
            if ch != "-":
                if first[i] == -1:
                    first[i] = j
                last[i] = j
    idx = {"A": 0, "a": 0, "C": 1, "c": 1, "G": 2, "g": 2, "T": 3, "t": 3, "N": 4, "n": 4, "-": 5}
    for j in range(ncol):
        s = 0
        for i, r in enumerate(rows):
            if first[i] <= j <= last[i]:
                s += 1
                if r[j] in idx:
                    p[idx[r[j]], j] += np.float32(1)
                else:
                    s -= 1
        if s > 0:
            p[:, j] = p[:, j] / np.float32(s)
    return p


def _cdiv(a, b):
    """C++ integer division (truncation toward zero)"""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def upgma(d, num):
    dim = 2 * num + 1
    p = [[-1, -1, -1] for _ in range(dim)]
    nn = num
    while nn < dim:
        best, dI, dJ = -1, 0, 0
        for i in range(nn):
            for j in range(i + 1, nn):
                if d[i][j] > best:
                    best, dI, dJ = d[i][j], i, j
        if best == -1:
            break
        p[dI][0] = p[dJ][0] = nn
        p[nn][1], p[nn][2] = dI, dJ
        for i in range(nn):
            if p[i][0] == -1:
                a = d[dI][i] if dI < i else d[i][dI]
                b = d[dJ][i] if dJ < i else d[i][dJ]
                d[i][nn] = _cdiv(a + b, 2)
        for i in range(dI):
            d[i][dI] = -1
        for i in range(dI + 1, nn + 1):
            d[dI][i] = -1
        for i in range(dJ):
            d[i][dJ] = -1
        for i in range(dJ + 1, nn + 1):
            d[dJ][i] = -1
        nn += 1
    return (nn - 1 if nn > 0 else 0), p


def palign(sps, p, root, score):
    if p[root][1] == -1 and p[root][2] == -1:
        prof = sps[root]
        return ["".join(cons_char(prof, j) for j in range(prof.shape[1]))], prof, [root]
    a1, p1, s1 = palign(sps, p, p[root][1], score)
    a2, p2, s2 = palign(sps, p, p[root][2], score)
    _, btr = orc.gotoh_prof(np.ascontiguousarray(p1), np.ascontiguousarray(p2), 1, 1, score)
    ops = btr[::-1].decode()  # forward order
    rows = [[] for _ in range(len(a1) + len(a2))]
    x = y = 0
    for op in ops:
        if op != "h":
            for k in range(len(a1)):
                rows[k].append(a1[k][x])
            x += 1
        else:
            for k in range(len(a1)):
                rows[k].append("-")
        if op != "v":
            for k in range(len(a2)):
                rows[len(a1) + k].append(a2[k][y])
            y += 1
        else:
            for k in range(len(a2)):
                rows[len(a1) + k].append("-")
    rows = ["".join(r) for r in rows]
    return rows, profile_of_alignment(rows), s1 + s2


def msa(sps, score):
    num = len(sps)
    dim = 2 * num + 1
    d = [[-1] * dim for _ in range(dim)]
    for i in range(num):
        for j in range(i + 1, num):
            d[i][j] = orc.gotoh_score_prof(sps[i], sps[j], 1, 1, score)
    root, p = upgma(d, num)
    rows, _, sidx = palign(sps, p, root, score)
    return rows, sidx


def consensus(rows, fraction_called=0.5, ignore_last=False):
    """msa.h:165-254 -> (gapped, cs, qstr)"""
    n = len(rows) - (1 if ignore_last else 0)
    ncol = len(rows[0])
    cov = [0] * ncol
    fl = [[False] * ncol for _ in range(n)]
    for i in range(n):
        start, end = 0, -1
        for j in range(ncol):
            if rows[i][j] != "-":
                end = j
            elif end == -1:
                start = j + 1
        for j in range(start, end + 1):
            cov[j] += 1
            fl[i][j] = True
    thr = int(np.float32(fraction_called) * np.float32(n))
    cons, qual, qualval = ["-"] * ncol, ["#"] * ncol, 33
    for j in range(ncol):
        max_idx = 4
        if cov[j] >= 1 and cov[j] >= thr:
            count = [0] * 5
            for i in range(n):
                if fl[i][j]:
                    c = rows[i][j].upper()
                    count["ACGT".index(c) if c in "ACGT" else 4] += 1
            max_idx, max_count = 0, count[0]
            for k in range(1, 5):
                if count[k] > max_count:
                    max_count, max_idx = count[k], k
            qualval = 47 + _cdiv(max_count * 10, n)
        if max_idx < 4:
            cons[j] = "ACGT"[max_idx]
            qual[j] = chr(qualval)
    gapped = "".join(cons)
    cs = "".join(c for c in cons if c != "-")
    qs = "".join(q for c, q in zip(cons, qual) if c != "-")
    return gapped, cs, qs


def rev_seq_based_on_dist(seq, score):
    """msa.h:258-323 -> (profiles, fwd flags)"""
    seq = [s.copy() for s in seq]
    num = len(seq)
    fwd = [True] * num
    d = [[0] * num for _ in range(num)]
    total = 0
    for i in range(num):
        for j in range(i + 1, num):
            d[i][j] = d[j][i] = orc.gotoh_score_prof(seq[i], seq[j], 1, 1, score)
            total += d[i][j]
    while True:
        quality = sorted((sum(d[i]), i) for i in range(num))
        for _, who in quality:
            s = orc.revcomp_profile(seq[who])
            new = [0] * num
            ssum = old = 0
            for i in range(num):
                if i != who:
                    new[i] = orc.gotoh_score_prof(seq[i], s, 1, 1, score)
                    old += d[i][who]
                    ssum += new[i]
            if ssum >= old:
                seq[who] = s
                fwd[who] = not fwd[who]
                for i in range(num):
                    d[i][who] = d[who][i] = new[i]
        updated = sum(sum(r) for r in d)
        if total < updated:
            total = updated
        else:
            break
    return seq, fwd
