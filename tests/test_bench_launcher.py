"""bench.py --gpus N must really run N ranks: without a torch.distributed environment it starts them itself, every rank
checks the size of its process group, and a box with fewer GPUs is refused.  CPU test of that launcher path on the gloo
backend with the stub step (`--stub`: no device work, the printed line says so)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def last_json(text):
    for ln in reversed(text.strip().split("\n")):
        if ln.startswith("{"):
            return json.loads(ln)
    raise AssertionError("no JSON line in: " + text[-500:])


def test_gpus_2_spawns_two_ranks():
    r = run(["--gpus", "2", "--stub", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = last_json(r.stdout)
    assert line["stub"] is True and line["n_gpus"] == 2 and line["backend"] == "gloo" and line["rccl_ranks"] == 0
    assert line["rank_sum"] == 3.0  # ranks 0 and 1 both took part in the collective
    assert line["steps"] == 3 and line["warmup"] == 1


def test_gpus_1_stub_is_one_rank():
    r = run(["--gpus", "1", "--stub", "--steps", "2", "--warmup", "0"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_PORT": "29533"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert last_json(r.stdout)["n_gpus"] == 1


def test_more_gpus_than_devices_is_refused():
    import torch
    have = torch.cuda.device_count()
    r = run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "--gpus %d" % (have + 2) in (r.stderr + r.stdout)
    assert "{" not in r.stdout  # no line, certainly none claiming n_gpus


def test_world_size_mismatch_is_refused():
    r = run(["--gpus", "4", "--stub"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_PORT": "29534"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
