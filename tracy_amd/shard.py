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


def gather_records(dist, records, dst=0):
    """Gather per-trace fixed-size records (tensor [n_local, F]) to `dst`, in rank order.
    Shards may differ in size by one trace; they are padded to the largest shard for the collective."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    n_local = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes) if sizes else 0
    padded = torch.zeros((cap,) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
    padded[:records.shape[0]] = records
    bucket = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bucket, dst=dst)
    if rank != dst:
        return None
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
