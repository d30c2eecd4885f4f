"""Generate tests/golden/trace_io.json from the REFERENCE's own abif.h (oracle/_ref/libref_abif.so): small
ABIF files written by the build's writer together with what the reference's readab() parsed from them, and
basecall() quality estimates.  Run in the build container only; the JSON is committed (data, not source)."""
import base64
import ctypes as C
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as orc  # noqa: E402
from test_trace_io import abif_cases, ref_read_trace  # noqa: E402
from test_host_and_abi import make_trace  # noqa: E402


def main():
    ref = orc.ref_lib()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(2027)
    out = {"abif": [], "estqual": []}
    with tempfile.TemporaryDirectory() as tmp:
        for p in abif_cases(rng, tmp)[:4]:
            t = ref_read_trace(p)
            out["abif"].append(dict(file_b64=base64.b64encode(open(p, "rb").read()).decode(), signal=t["signal"].tolist(),
                                    basecallpos=t["basecallpos"].tolist(), basecalls1_b64=base64.b64encode(t["basecalls1"]).decode(),
                                    basecalls2_b64=base64.b64encode(t["basecalls2"]).decode(), qual=t["qual"].tolist()))
    ref.ref_basecall_qual.restype = C.c_size_t
    for nb in [1, 9, 11, 30, 64, 120, 21]:
        tr, pos = make_trace(rng, nb, het=[0.0, 0.5][nb % 2])
        n = len(pos)
        pri, sec, con = (C.create_string_buffer(n + 1) for _ in range(3))
        bc = np.zeros(max(n, 1), np.int32)
        q = np.zeros(max(n, 1), np.uint8)
        k = ref.ref_basecall_qual(tr.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(tr.shape[1]), pos.ctypes.data_as(C.POINTER(C.c_int32)),
                                  C.c_size_t(n), C.c_float(0.33), pri, sec, con, bc.ctypes.data_as(C.POINTER(C.c_int32)),
                                  q.ctypes.data_as(C.POINTER(C.c_uint8)))
        out["estqual"].append(dict(trace=tr.tolist(), basecallpos=pos.tolist(), sigratio=0.33, primary=pri.raw[:k].decode(),
                                   estQual=q[:k].tolist()))
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "trace_io.json"), "w"))
    print("wrote", len(out["abif"]), "ABIF files and", len(out["estqual"]), "quality cases")


if __name__ == "__main__":
    main()
