"""Generate tests/golden/abif_basecall.json and abif_iupac.json from the REFERENCE's own abif.h
(oracle/_ref/libref_abif.so, built by oracle/Makefile from /root/reference/src/abif.h).  Run in the
build container only (the reference does not exist on the GPU box); the JSON files are committed."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as orc  # noqa: E402
from test_host_and_abi import make_trace  # noqa: E402


def main():
    ref = orc.ref_lib()
    assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.default_rng(2026)
    cases = []
    for it in range(8):
        nb = [12, 40, 25, 60, 33, 80, 18, 50][it]
        tr, pos = make_trace(rng, nb, het=[0.0, 0.4, 0.9, 0.2][it % 4])
        if it == 5:
            tr[:, 200:260] = 0          # a dead stretch: no peaks, midpoint fallback
        if it == 6:
            pos[3] = pos[2]             # a zero-width window: peak() rejects it (abif.h:81)
        sig = [0.33, 0.33, 0.2, 0.5][it % 4]
        pri, sec, con, bcpos = orc.ref_basecall(tr, pos, sig)
        cases.append(dict(trace=tr.tolist(), basecallpos=pos.tolist(), sigratio=sig, primary=pri.decode(),
                          secondary=sec.decode(), consensus=con.decode(), bcPos=bcpos.tolist()))
    json.dump(cases, open(os.path.join(ROOT, "tests", "golden", "abif_basecall.json"), "w"))
    iu = []
    for a in "ACGTNR":
        for b in "ACGTNY":
            iu.append([a, b, ref.ref_iupac2(a.encode(), b.encode()).decode()])
    json.dump(iu, open(os.path.join(ROOT, "tests", "golden", "abif_iupac.json"), "w"))
    print("wrote", len(cases), "basecall cases and", len(iu), "iupac pairs")


if __name__ == "__main__":
    main()
