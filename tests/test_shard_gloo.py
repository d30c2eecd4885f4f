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
    # the same with the shard sizes handed in (no size exchange): what bench.py's steps do
    sizes = [b - a for a, b in (shard_range(n, r, world) for r in range(world))]
    again = gather_records(dist, rec_t, dst=0, sizes=sizes)
    assert rank != 0 or torch.equal(again, all_rec)
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


# ---- both halves of the final gather as bench.py's steps run them (ResultGather): records in one collective, the ragged payloads -- two
# kinds here, bytes and int32 rows -- packed per rank and shipped in one exchange sized from the records' length columns ----
def _result_worker(rank, world, port, n, ret):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tracy_amd.shard import ResultGather, shard_range, unpack_ragged
    qs, refs = _make_batch(n)
    lo, hi = shard_range(n, rank, world)
    recs, ops = _align(qs[lo:hi], refs[lo:hi])
    nl, cap, tcap = hi - lo, 300, 7
    buf = torch.zeros(nl * cap, dtype=torch.uint8)        # traceback strings in fixed-capacity regions (ops_offset[i] = i * cap)
    tab = torch.full((nl * tcap,), -1, dtype=torch.int32)  # a table of int32 rows per trace (the decomposition table's shape)
    rec = torch.zeros((nl, 4), dtype=torch.int32)
    for i, ((sc, ln), o) in enumerate(zip(recs, ops)):
        buf[i * cap:i * cap + ln] = torch.from_numpy(np.frombuffer(o, dtype=np.uint8).copy())
        rows = (lo + i) % (tcap + 1)
        tab[i * tcap:i * tcap + rows] = torch.arange(rows, dtype=torch.int32) + 100 * (lo + i)
        rec[i] = torch.tensor([sc, ln, rows, lo + i], dtype=torch.int32)
    sizes = [b - a for a, b in (shard_range(n, r, world) for r in range(world))]
    g = ResultGather(dist, sizes)
    pay = [(buf, cap, 1), (tab, tcap, 2)]
    allrec, got = g.gather(rec, pay)
    if rank == 0:
        assert g.check_own_block(allrec, got, rec, pay)
        ret["rec"] = allrec.numpy().tolist()
        ret["ops"] = unpack_ragged(got[0], allrec[:, 1])
        ret["tab"] = [np.frombuffer(b, dtype=np.int32).tolist() for b in unpack_ragged(got[1], allrec[:, 2], 4)]
    else:
        assert allrec is None and got is None
    ret["bytes%d" % rank] = g.bytes_last
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_result_gather_both_halves():
    n = 9
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_result_worker, args=(2, port, n, ret), nprocs=2, join=True)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    qs, refs = _make_batch(n)
    recs, ops = _align(qs, refs)
    assert [r[:2] for r in ret["rec"]] == recs and [r[3] for r in ret["rec"]] == list(range(n))
    assert ret["ops"] == ops  # every trace's string, the second rank's included, as one process computes them
    assert ret["tab"] == [[100 * i + k for k in range(i % 8)] for i in range(n)]
    assert ret["bytes0"] + ret["bytes1"] == 16 * n + sum(len(o) for o in ops) + 4 * sum(i % 8 for i in range(n))


def test_shard_range_covers_batch():
    from tracy_amd.shard import shard_range
    for n in (0, 1, 7, 8, 100001):
        for w in (1, 2, 3, 8):
            got = [shard_range(n, r, w) for r in range(w)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            assert max(h - l for l, h in got) - min(h - l for l, h in got) <= 1


# ---- all-pairs jobs (`tracy assemble`, msa.h:33-42): the pair list sharded over ranks, score slices all-gathered -----------
def _profiles(n):
    rng = np.random.default_rng(77)
    out = []
    for _ in range(n):
        m = int(rng.integers(40, 120))  # ragged: slices of equal CELL count differ in pair count
        p = np.zeros((6, m), np.float32)
        x = rng.random((4, m)).astype(np.float32) ** 4
        p[:4] = x / x.sum(axis=0, keepdims=True)
        out.append(p)
    return out


def _allpairs_worker(rank, world, port, n, ret):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyoracle as orc
    from tracy_amd.shard import all_gather_slices, pair_slice
    profs = _profiles(n)
    lens = [p.shape[1] for p in profs]
    i1, i2, bounds = pair_slice(lens, rank, world)
    mine = torch.tensor([orc.gotoh_score_prof(profs[a], profs[b], 1, 1, SC) for a, b in zip(i1, i2)], dtype=torch.int32)  # oracle = the device here
    full = all_gather_slices(dist, mine, bounds)
    ret["full%d" % rank] = full.numpy().tolist()
    ret["bounds%d" % rank] = [int(b) for b in bounds]
    ret["idx%d" % rank] = (i1.tolist(), i2.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_all_pairs_matrix():
    """a 40-trace all-pairs job: ranks take contiguous slices of msa()'s own pair list (equal cell counts), every rank ends up
    with the whole distance matrix, identical to the single-process one"""
    n = 40
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_allpairs_worker, args=(2, port, n, ret), nprocs=2, join=True)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    from tracy_amd import msalib
    from tracy_amd.shard import condensed_to_square
    profs = _profiles(n)
    m1, m2 = msalib.pair_list(n)  # the index arrays msa.hpp hands to tracyhip_gotoh_score
    want = [orc.gotoh_score_prof(profs[a], profs[b], 1, 1, SC) for a, b in zip(m1, m2)]
    assert ret["full0"] == want and ret["full1"] == want
    assert ret["bounds0"] == ret["bounds1"] and ret["bounds0"][0] == 0 and ret["bounds0"][-1] == n * (n - 1) // 2
    # the ranks' index arrays, concatenated, are exactly msa()'s pair list
    assert ret["idx0"][0] + ret["idx1"][0] == m1.tolist() and ret["idx0"][1] + ret["idx1"][1] == m2.tolist()
    # slices carry (nearly) equal numbers of DP cells, not of pairs
    lens = np.array([p.shape[1] for p in profs], dtype=np.int64)
    cells = lens[m1] * lens[m2]
    b = ret["bounds0"]
    c0, c1 = int(cells[:b[1]].sum()), int(cells[b[1]:].sum())
    assert abs(c0 - c1) <= int(cells.max())
    sq = condensed_to_square(want, n)
    assert sq.shape == (n, n) and (sq == sq.T).all() and sq[3, 17] == want[m1.tolist().index(3) + 17 - 3 - 1]


def test_pair_bounds_rule():
    from tracy_amd.shard import pair_bounds
    for n, w in ((2, 1), (2, 2), (5, 3), (40, 8), (100, 7)):
        lens = np.arange(50, 50 + n)
        b = pair_bounds(lens, w)
        assert b[0] == 0 and b[-1] == n * (n - 1) // 2 and (np.diff(b) >= 0).all()


def _seed_worker(rank, world, port, tmp, ret):
    """the host side of configs[3] on two ranks: thread share per rank, one index file built by rank 0 and mapped by both"""
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    os.environ["WORLD_SIZE"] = str(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bench import usable_cores
    from tools.legs import rank_threads
    from tracy_amd import hostlib
    from tracy_amd.shard import shard_range
    threads = rank_threads(world)
    rng = np.random.default_rng(4)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=60000)].tobytes()
    fa, idx = os.path.join(tmp, "g.fa"), os.path.join(tmp, "g.tidx")
    if rank == 0:
        open(fa, "wb").write(b">chrSyn\n" + seq + b"\n")
        g0 = hostlib.Genome(fa, 15, threads)
        g0.save(idx)
        g0.close()
    dist.barrier()
    g = hostlib.Genome(idx, 15, threads)  # mapped by every rank
    starts = rng.integers(0, 60000 - 400, size=21)
    reads = [seq[s:s + 400] for s in starts]
    lo, hi = shard_range(len(reads), rank, world)
    got = g.seed(reads[lo:hi], 50, 50, 3, 1000, threads)
    ret[rank] = dict(threads=threads, usable=usable_cores(), pos=[int(x) for x in got["pos"]], status=[int(x) for x in got["status"]], lo=lo, hi=hi,
                     starts=[int(x) for x in starts])
    g.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_one_mapped_index_and_split_the_host_threads(tmp_path):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_seed_worker, args=(2, port, str(tmp_path), ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["threads"] == r1["threads"] == max(1, r0["usable"] // 2)
    assert (r0["lo"], r0["hi"], r1["hi"]) == (0, r1["lo"], 21)
    for r in (r0, r1):
        assert all(s == 1 for s in r["status"])
        for k, p in enumerate(r["pos"]):  # window start = locus - maxindel, clipped at the contig start
            assert p == max(0, r["starts"][r["lo"] + k] - 1000)
