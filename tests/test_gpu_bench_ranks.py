"""The N-rank code path of bench.py on a one-GPU box: `--gpus 2 --share-device` starts two ranks (torch.distributed.run, one process
each) that share GPU 0 and gather over gloo.  Not a measurement (the line says "shared_device": true) -- a test that a multi-GPU run
prints ONE line from rank 0 with everything the single-GPU line carries: roofline, cpu_baseline (rank 0, its share of the cores),
the rank count, the per-rank spread, and that the decompose leg shards one job over the ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 alone prints
    return json.loads(lines[0])


def test_two_ranks_headline_line():
    line = run(["--gpus", "2", "--share-device", "--workload", "align", "--traces", "384", "--ref-len", "3000", "--steps", "2", "--warmup", "1",
                "--cpu-sample", "8", "--lanes-leg", "0"])
    assert line["n_gpus"] == 2 and line["backend"] == "gloo" and line["rccl_ranks"] == 0 and line["shared_device"] is True  # (two ranks on one GPU: gloo, RCCL saw none)
    assert line["scaling"] == "weak" and line["config"]["traces_per_gpu"] == 384
    assert line["traces_per_s"] > 0 and line["value"] > 0
    r = line["roofline"]
    assert r["bound"] == "valu" and r["frac"] > 0 and r["hbm_achieved_gbs"] > 0 and "traffic" in r and r["dominant_kernel_alone_frac"] > 0
    # both halves of the gather ran in the timed steps: 9 int32 per trace + the traceback strings of BOTH ranks reached rank 0, its own block checked
    assert line["gather_checked"] is True and line["gathered_bytes_per_step"] > 2 * 384 * (9 * 4 + 900)
    assert len(json.dumps(line)) < 8000
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    assert line["parity_checked"]["bit_identical"] is True
    p = line["pipeline"]
    assert p["stream_ordered"] == 1 and p["host_syncs_per_call"] <= 2
    assert 0 < p["ms_per_step_rank_min"] <= p["ms_per_step_rank_max"]
    assert line["strand_by_certificate"]["alignments_identical_to_headline_leg"] is True


def test_two_ranks_decompose_leg_shards_one_job():
    line = run(["--gpus", "2", "--share-device", "--workload", "decompose", "--decompose-traces", "600", "--decompose-steps", "2", "--extra-legs", "0"])
    assert line["n_gpus"] == 2 and line["backend"] == "gloo" and line["rccl_ranks"] == 0 and line["scaling"] == "strong"
    assert line["config"]["traces_total"] == 600 and line["pipeline"]["traces_per_rank"] == 300
    assert line["pipeline"]["stream_ordered"] == 1
    assert line["cpu_baseline"]["value"] > 0 and line["parity_checked"]["bit_identical"] is True
    assert line["roofline"]["bound"] in ("valu", "hbm")
    # records + three traceback strings + rewritten basecalls + secDecompose + decomposition tables of both ranks
    assert line["gather_checked"] is True and line["gathered_bytes_per_step"] > 600 * (32 * 4 + 3 * 1000 + 3 * 900)
