"""A slice of tools/parity_100k.py (the north star's parity sentence: every trace of the two headline batches against the
oracle) inside the GPU suite, drawn from ALL OVER the two batches: eight blocks of 256 traces at seeds spread over the 10 000
traces of the configs[1] batch and over the 100 000 of the configs[2] batch, through the same code -- all fields of every trace,
the certificate / two-lane modes against the exact one-lane mode on the device -- plus twenty rounds of the randomized campaign
(tests/fuzz_parity.py: every DP mode / AlignConfig / scoring, ragged pipelines, trims / maxindel / MAD cut-off varied)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BLOCK = 256
ALIGN_STARTS = [0, 1218, 2436, 3654, 4872, 6090, 7308, 10000 - BLOCK]           # (bench.py: trace i of the batch has seed 1000 + i)
DECOMPOSE_STARTS = [0, 14250, 28500, 42750, 57000, 71250, 85500, 100000 - BLOCK]  # (tools/legs.py: seed 5000 + i)


@pytest.mark.parametrize("first", ALIGN_STARTS)
def test_align_batch_block_every_trace(first):
    from tools import parity_100k as P
    log = P.Log(None)
    compared, mism = P.run_align(BLOCK, 10000, 1000, log, block=128, seed=1000 + first, nthreads=16)
    assert compared == BLOCK and mism == 0, [ln for ln in log.lines if ln.get("mismatches") or ln.get("modes")]
    modes = next(ln for ln in log.lines if "modes" in ln)["modes"]
    assert set(modes) == {"certificate_1lane", "exact_2lanes", "certificate_2lanes"}


@pytest.mark.parametrize("first", DECOMPOSE_STARTS)
def test_decompose_batch_block_every_trace(first):
    from tools import parity_100k as P
    log = P.Log(None)
    compared, mism, accepted = P.run_decompose(BLOCK, 3000, 1000, log, block=128, nthreads=16, first=first)
    assert compared == BLOCK and mism == 0, [ln for ln in log.lines if ln.get("mismatches") or ln.get("modes")]
    assert accepted >= 200  # the chain accepts almost every synthetic trace: the deep fields were compared


def test_twenty_rounds_of_the_randomized_campaign():
    from fuzz_parity import run_campaign
    scorings = set()
    for r in range(20):
        res = run_campaign(pairs=900, traces=64, lanes=2 if r % 2 else 1, seed=20260500 + r)
        assert res["mismatches"] == 0, res
        scorings.add(tuple(res["pipeline_scoring"]))
    assert len(scorings) >= 3
