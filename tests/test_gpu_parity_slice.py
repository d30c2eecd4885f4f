"""A slice of tools/parity_100k.py (the north star's parity sentence: every trace of the two headline batches against the
oracle) inside the GPU suite: 256 traces of the configs[1] batch and 256 of the configs[2] batch, through the same code --
all fields of every trace, the certificate / two-lane modes against the exact one-lane mode on the device."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_align_batch_slice_every_trace():
    from tools import parity_100k as P
    log = P.Log(None)
    compared, mism = P.run_align(256, 10000, 1000, log, block=128, nthreads=16)
    assert compared == 256 and mism == 0, [ln for ln in log.lines if ln.get("mismatches") or ln.get("modes")]
    modes = next(ln for ln in log.lines if "modes" in ln)["modes"]
    assert set(modes) == {"certificate_1lane", "exact_2lanes", "certificate_2lanes"}


def test_decompose_batch_slice_every_trace():
    from tools import parity_100k as P
    log = P.Log(None)
    compared, mism, accepted = P.run_decompose(256, 3000, 1000, log, block=128, nthreads=16)
    assert compared == 256 and mism == 0, [ln for ln in log.lines if ln.get("mismatches") or ln.get("modes")]
    assert accepted >= 200  # the chain accepts almost every synthetic trace: the deep fields were compared
