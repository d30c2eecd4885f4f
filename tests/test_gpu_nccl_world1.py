"""The RCCL path once, at world size 1, on the GPU (the driver's 8-GPU run is one shot): bench.py started the way torch.distributed.run
starts a rank (RANK / WORLD_SIZE / LOCAL_RANK in the environment -> init_process_group("nccl")), and the gather helpers of
tracy_amd/shard.py on CUDA tensors through the nccl backend."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env():
    env = dict(os.environ)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_bench_rank_under_a_torchrun_environment_uses_nccl():
    for args, check in (
        (["--workload", "align", "--traces", "512", "--ref-len", "3000", "--steps", "2", "--warmup", "1", "--cpu-sample", "8", "--lanes-leg", "0", "--certificate-leg", "0"],
         lambda ln: ln["config"]["traces_per_gpu"] == 512 and ln["value"] > 0 and ln["config"]["gcups_swept_cells"] > 0 and ln["gather_checked"] is True
         and ln["gathered_bytes_per_step"] > 512 * 900),
        (["--workload", "decompose", "--decompose-traces", "400", "--decompose-steps", "2", "--extra-legs", "0", "--cpu-sample", "0"],
         lambda ln: ln["config"]["traces_total"] == 400 and ln["pipeline"]["traces_per_rank"] == 400 and ln["gather_checked"] is True
         and ln["gathered_bytes_per_step"] > 400 * 5000),
        (["--workload", "allpairs", "--allpairs-traces", "64", "--allpairs-steps", "1", "--cpu-sample", "0"], lambda ln: ln["value"] > 0),
    ):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, capture_output=True, text=True, env=rank_env(), timeout=900, cwd=ROOT)
        assert r.returncode == 0, (r.stdout[-800:], r.stderr[-3000:])
        lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        line = json.loads(lines[0])
        assert line["n_gpus"] == 1 and line["backend"] == "nccl" and line["rccl_ranks"] == 1, {k: line.get(k) for k in ("n_gpus", "backend", "rccl_ranks")}
        assert check(line), line


HELPERS = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
import torch
import torch.distributed as dist
from tracy_amd import shard
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
rec = torch.arange(7 * 5, dtype=torch.int32, device="cuda").reshape(7, 5)
got = shard.gather_records(dist, rec)
assert got.is_cuda and torch.equal(got, rec)
lens = torch.tensor([3, 0, 5, 1], dtype=torch.int64, device="cuda")
data = torch.arange(9, dtype=torch.uint8, device="cuda")
d2, l2 = shard.gather_ragged_bytes(dist, data, lens)
assert torch.equal(d2, data) and torch.equal(l2, lens)
lengths = np.array([900, 700, 800, 650, 720], dtype=np.uint32)
i1, i2, b = shard.pair_slice(lengths, 0, 1)
local = torch.arange(len(i1), dtype=torch.int32, device="cuda")
# both halves of the result gather on CUDA tensors: tracyhip_pack_ragged packs, RCCL ships
import tracy_amd
ctx = tracy_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n, cap = 5, 64
lens_h = [3, 0, 64, 17, 1]
buf = torch.arange(n * cap, dtype=torch.int64, device="cuda").to(torch.uint8)
recs = torch.tensor([[7 * i, lens_h[i]] for i in range(n)], dtype=torch.int32, device="cuda")
g = shard.ResultGather(dist, [n], ctx)
allrec, got = g.gather(recs, [(buf, cap, 1)])
want = torch.cat([buf[i * cap:i * cap + lens_h[i]] for i in range(n)])
assert torch.equal(allrec, recs) and torch.equal(got[0], want) and g.bytes_last == 8 * n + sum(lens_h)
assert g.check_own_block(allrec, got, recs, [(buf, cap, 1)])
tab = torch.arange(n * 4, dtype=torch.int32, device="cuda")
rows = torch.tensor([[0, 4, 2, 1, 3][i] for i in range(n)], dtype=torch.int32, device="cuda").reshape(n, 1)
p4, nb = ctx.pack_ragged(tab, 4, rows)
assert nb == 40 and torch.equal(p4.view(torch.int32), torch.cat([tab[i * 4:i * 4 + int(rows[i])] for i in range(n)]))
ctx.close()
allv = shard.all_gather_slices(dist, local, b)
assert torch.equal(allv, local)
t = torch.tensor([1.5, 2.0], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("nccl helpers ok")
'''


def test_shard_helpers_on_cuda_tensors_through_nccl():
    r = subprocess.run([sys.executable, "-c", HELPERS % ROOT], capture_output=True, text=True, env=rank_env(), timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "nccl helpers ok" in r.stdout, (r.stdout[-800:], r.stderr[-3000:])
