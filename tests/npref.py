"""Independent full-matrix formulation of the Gotoh / NW recurrences, written
from the matrix view (H/E/F planes + explicit bit planes) rather than the reference's rolling rows.
Pure Python: use only for small cases.  Used to cross-check the C oracle (tests only)."""
import numpy as np

NEG = -1000000


def sub_str(s1, s2, r, c, sc):
    return sc[0] if s1[r] == s2[c] else sc[1]


def sub_prof(p1, p2, r, c, sc):
    acc = np.float32(0)
    for k1 in range(5):
        for k2 in range(5):
            w = np.float32(sc[0] if k1 == k2 else sc[1])
            acc = np.float32(acc + np.float32(np.float32(p1[k1, r] * p2[k2, c]) * w))
    return int(acc)  # truncation toward zero


def gotoh_full(m, n, sub, hfree, vfree, sc):
    match, mismatch, go, ge = sc
    H = np.zeros((m + 1, n + 1), dtype=np.int64)
    E = np.full((m + 1, n + 1), NEG, dtype=np.int64)
    F = np.full((m + 1, n + 1), NEG, dtype=np.int64)
    b1 = np.zeros((m + 1, n + 1), dtype=bool)
    b2 = np.zeros_like(b1)
    b3 = np.zeros_like(b1)
    b4 = np.zeros_like(b1)
    b1[0, 0] = b2[0, 0] = True
    for c in range(1, n + 1):
        H[0, c] = E[0, c] = 0 if hfree else go + c * ge
        b3[0, c] = True
    for r in range(1, m + 1):
        H[r, 0] = F[r, 0] = 0 if vfree else go + r * ge
        b4[r, 0] = True
    for r in range(1, m + 1):
        hz = hfree and r == m
        for c in range(1, n + 1):
            vz = vfree and c == n
            eo = H[r, c - 1] + (0 if hz else go + ge)
            ee = E[r, c - 1] + (0 if hz else ge)
            E[r, c] = max(eo, ee)
            fo = H[r - 1, c] + (0 if vz else go + ge)
            fe = F[r - 1, c] + (0 if vz else ge)
            F[r, c] = max(fo, fe)
            H[r, c] = max(H[r - 1, c - 1] + sub(r - 1, c - 1), E[r, c], F[r, c])
            b3[r, c] = H[r, c] == E[r, c]
            b4[r, c] = (not b3[r, c]) and H[r, c] == F[r, c]
            b1[r, c] = E[r, c] != ee
            b2[r, c] = F[r, c] != fe
    # traceback state machine (gotoh.h:143-167 semantics)
    r, c, st, ops = m, n, "s", []
    while r > 0 or c > 0:
        if st == "s":
            if b3[r, c]:
                st = "h"
            elif b4[r, c]:
                st = "v"
            else:
                r -= 1
                c -= 1
                ops.append("s")
        elif st == "h":
            if b1[r, c]:
                st = "s"
            c -= 1
            ops.append("h")
        else:
            if b2[r, c]:
                st = "s"
            r -= 1
            ops.append("v")
    return int(H[m, n]), "".join(ops).encode()


def needle_full(m, n, sub, hfree, vfree, sc):
    match, mismatch, go, ge = sc
    H = np.zeros((m + 1, n + 1), dtype=np.int64)
    b3 = np.zeros((m + 1, n + 1), dtype=bool)
    b4 = np.zeros_like(b3)
    for c in range(1, n + 1):
        H[0, c] = 0 if hfree else c * ge
        b3[0, c] = True
    for r in range(1, m + 1):
        H[r, 0] = 0 if vfree else r * ge
        b4[r, 0] = True
    for r in range(1, m + 1):
        hz = hfree and r == m
        for c in range(1, n + 1):
            vz = vfree and c == n
            hor = H[r, c - 1] + (0 if hz else ge)
            ver = H[r - 1, c] + (0 if vz else ge)
            H[r, c] = max(H[r - 1, c - 1] + sub(r - 1, c - 1), ver, hor)
            b3[r, c] = H[r, c] == hor
            b4[r, c] = (not b3[r, c]) and H[r, c] == ver
    r, c, ops = m, n, []
    while r > 0 or c > 0:
        if b3[r, c]:
            c -= 1
            ops.append("h")
        elif b4[r, c]:
            r -= 1
            ops.append("v")
        else:
            r -= 1
            c -= 1
            ops.append("s")
    return int(H[m, n]), "".join(ops).encode()


def sub_prof_double(p1, p2, r, c, sc):
    """needle.h: double profiles, float accumulator"""
    acc = np.float32(0)
    for k1 in range(5):
        for k2 in range(5):
            w = float(sc[0] if k1 == k2 else sc[1])
            acc = np.float32(float(acc) + float(p1[k1, r]) * float(p2[k2, c]) * w)
    return int(acc)
