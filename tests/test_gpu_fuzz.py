"""A bounded slice of the randomized GPU-vs-oracle campaign (tests/fuzz_parity.py) inside `pytest -m gpu`: fixed seeds,
one lane and two lanes; every DP mode / AlignConfig / scoring on ragged pairs, ragged `tracy align` batches in both
orientation modes, `tracy decompose` batches (het indel / homozygous / none, both strands).  The full campaign's log of
the round is committed under profiles/ (tools/profile_round.sh)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,lanes", [(20260101, 1), (20260102, 2)])
def test_fuzz_slice(seed, lanes):
    from fuzz_parity import run_campaign
    res = run_campaign(pairs=600, traces=160 if lanes > 1 else 48, lanes=lanes, seed=seed)
    assert res["mismatches"] == 0, res
    assert res["compared"]["dp_char"] > 0 and res["compared"]["align"] > 0 and res["compared"]["decompose"] > 0
