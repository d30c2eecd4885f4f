"""Batch sharding across the GPUs of one node.

Every trace x reference pair is independent, so a batch shards by index with no data-path collective;
the only communication is the final gather of the fixed-size result records (and, optionally, of the
variable-length traceback strings) to rank 0 -- RCCL over xGMI when the process group is `nccl`, gloo in
the CPU tests.  One process per GPU (torch.distributed)."""
import torch


def shard_range(n, rank, world):
    """contiguous block of the batch owned by `rank`: [lo, hi)"""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def gather_records(dist, records, dst=0, sizes=None):
    """Gather per-trace fixed-size records (tensor [n_local, F]) to `dst`, in rank order.
    Shards may differ in size by one trace; they are padded to the largest shard for the collective.
    sizes: the shard sizes of all ranks when the caller knows them (shard_range gives them without asking anyone): ONE collective and no
    host synchronisation; without them the sizes are all-gathered first."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    if records.is_cuda and dist.get_backend() == "gloo":  # (gloo gathers host tensors: the CPU tests, bench.py --share-device)
        records = records.cpu()
    if sizes is None:
        n_local = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
        sizes = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(sizes, n_local)
        sizes = [int(s.item()) for s in sizes]
    else:
        sizes = [int(x) for x in sizes]
        if len(sizes) != world or sizes[rank] != records.shape[0]:
            raise ValueError("gather_records: sizes %r do not describe this rank's %d records" % (sizes, records.shape[0]))
    cap = max(sizes) if sizes else 0
    if records.shape[0] == cap and records.is_contiguous():
        padded = records
    else:
        padded = torch.zeros((cap,) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
        padded[:records.shape[0]] = records
    bucket = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bucket, dst=dst)
    if rank != dst:
        return None
    if world == 1:
        return bucket[0]
    return torch.cat([b[:s] for b, s in zip(bucket, sizes)], dim=0)


def gather_ragged_bytes(dist, data, lengths, dst=0):
    """Gather variable-length byte strings (e.g. traceback ops): `data` uint8 [sum(lengths)], `lengths`
    int64 [n_local].  Returns (data, lengths) concatenated in rank order on `dst`."""
    lens = gather_records(dist, lengths.reshape(-1, 1), dst)
    total = torch.tensor([[data.numel()]], dtype=torch.int64, device=data.device)
    totals = gather_records(dist, total, dst)
    world = dist.get_world_size()
    rank = dist.get_rank()
    all_tot = [torch.zeros(1, dtype=torch.int64, device=data.device) for _ in range(world)]
    dist.all_gather(all_tot, total.reshape(1))
    cap = max(int(t.item()) for t in all_tot)
    padded = torch.zeros(cap, dtype=torch.uint8, device=data.device)
    padded[:data.numel()] = data
    bucket = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bucket, dst=dst)
    if rank != dst:
        return None, None
    out = torch.cat([b[:int(t.item())] for b, t in zip(bucket, all_tot)])
    return out, lens.reshape(-1)


# ---- all-pairs jobs (`tracy assemble`, msa.h:33-42 distanceMatrix) ---------------------------------------------------
def pair_bounds(lengths, world):
    """The upper-triangular pair list (i < j, row-major: the order of the two loops of msa.h:33-42) cut into `world`
    contiguous slices of (nearly) equal DP cell count len[i] * len[j].  Returns world + 1 boundaries into the pair list.
    The rule is the library's (tracyhip_pair_bounds, host arithmetic): device groups and ranks cut the list identically."""
    import numpy as np
    from . import capi
    lengths = np.asarray(lengths, dtype=np.uint32)
    i1, i2 = np.triu_indices(len(lengths), 1)
    return capi.pair_bounds(lengths, lengths, i1, i2, world)


def pair_slice(lengths, rank, world):
    """index arrays (a1_index, a2_index) of this rank's slice of the pair list + the boundaries of every rank's slice"""
    import numpy as np
    n = len(lengths)
    b = pair_bounds(lengths, world)
    i1, i2 = np.triu_indices(n, 1)
    lo, hi = int(b[rank]), int(b[rank + 1])
    return i1[lo:hi].astype(np.uint32), i2[lo:hi].astype(np.uint32), b


def all_gather_slices(dist, local, bounds):
    """all_gather of the ranks' score slices (one collective, slices padded to the largest): every rank receives the scores
    of the whole pair list in pair order -- the distance matrix of msa.h:33-42 in condensed form (<= 4 MB for 1000 traces)."""
    world = dist.get_world_size()
    sizes = [int(bounds[r + 1] - bounds[r]) for r in range(world)]
    cap = max(sizes) if sizes else 0
    if local.is_cuda and dist.get_backend() == "gloo":
        local = local.cpu()
    padded = torch.zeros(cap, dtype=local.dtype, device=local.device)
    padded[:local.numel()] = local
    bucket = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bucket, padded)
    return torch.cat([b[:s] for b, s in zip(bucket, sizes)])


def condensed_to_square(scores, n):
    """condensed pair-order scores -> symmetric n x n matrix with a zero diagonal (what msa.h:33-42 fills)"""
    import numpy as np
    m = np.zeros((n, n), dtype=np.asarray(scores).dtype)
    iu = np.triu_indices(n, 1)
    m[iu] = np.asarray(scores)
    return m + m.T
