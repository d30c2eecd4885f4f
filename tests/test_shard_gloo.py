"""N>1 path on CPU: world_size-2 gloo processes shard a batch by index, align their shard (the oracle
stands in for the GPU here -- tests only), gather the records to rank 0, and rank 0 checks the result
equals the single-process one."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SC = (3, -5, -10, -4)


def _make_batch(n):
    rng = np.random.default_rng(123)
    refs = [bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(80, 200))).tolist()) for _ in range(n)]
    qs = [r[10:10 + int(rng.integers(20, 60))] for r in refs]
    return qs, refs


def _align(qs, refs):
    import pyoracle as orc
    recs, ops = [], []
    for q, r in zip(qs, refs):
        s, btr = orc.gotoh_str(q, r, 1, 0, SC)
        recs.append([s, len(btr)])
        ops.append(btr)
    return recs, ops


def _worker(rank, world, port, n, ret):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tracy_amd.shard import gather_ragged_bytes, gather_records, shard_range
    qs, refs = _make_batch(n)
    lo, hi = shard_range(n, rank, world)
    recs, ops = _align(qs[lo:hi], refs[lo:hi])
    rec_t = torch.tensor(recs, dtype=torch.int64).reshape(-1, 2)
    all_rec = gather_records(dist, rec_t, dst=0)
    data = torch.from_numpy(np.frombuffer(b"".join(ops), dtype=np.uint8).copy())
    lens = torch.tensor([len(o) for o in ops], dtype=torch.int64)
    all_ops, all_lens = gather_ragged_bytes(dist, data, lens, dst=0)
    if rank == 0:
        ret["rec"] = all_rec.numpy().tolist()
        ret["ops"] = bytes(all_ops.numpy().tobytes())
        ret["lens"] = all_lens.numpy().tolist()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    n = 7  # odd: the two shards differ in size
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, n, ret), nprocs=2, join=True)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    qs, refs = _make_batch(n)
    recs, ops = _align(qs, refs)
    assert ret["rec"] == recs
    assert ret["ops"] == b"".join(ops)
    assert ret["lens"] == [len(o) for o in ops]


def test_shard_range_covers_batch():
    from tracy_amd.shard import shard_range
    for n in (0, 1, 7, 8, 100001):
        for w in (1, 2, 3, 8):
            got = [shard_range(n, r, w) for r in range(w)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            assert max(h - l for l, h in got) - min(h - l for l, h in got) <= 1
