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


def pack_ragged(buf, stride, lengths):
    """Pack the used part of fixed-capacity per-trace regions: trace i owns `stride` elements of `buf` (any dtype) from i * stride and
    uses the first lengths[i] of them (traceback strings at ops_offset[i] = i * cap; rows of a decomposition table).  Returns the used
    elements back to back as uint8 (trace order) and the total in BYTES (a Python int: the one host read this costs)."""
    n = int(lengths.numel())
    if n == 0:
        return torch.zeros(0, dtype=torch.uint8, device=buf.device), 0
    lens = lengths.to(torch.int64)
    width = int(lens.max().item())
    rows = buf[:n * stride].view(n, stride)[:, :width]
    keep = torch.arange(width, device=buf.device)[None, :] < lens[:, None]
    packed = rows[keep].contiguous().view(torch.uint8)
    return packed, int(packed.numel())


def unpack_ragged(packed, lengths, elem_bytes=1):
    """the strings pack_ragged / gather_ragged_known deliver, as a list of bytes objects (tests, the parity samples)"""
    data = packed.cpu().numpy().tobytes()
    out, at = [], 0
    for ln in lengths.cpu().numpy().astype("int64").tolist():
        out.append(data[at:at + ln * elem_bytes])
        at += ln * elem_bytes
    return out


def gather_ragged_known(dist, packed, totals, dst=0):
    """Second half of the result gather: variable-length payloads -- traceback strings, rewritten basecalls, decomposition
    tables -- packed back to back per rank (pack_ragged), to `dst` in rank order.  `totals`: every rank's byte count AS `dst` KNOWS IT
    from the fixed-size records it gathered first (their length columns); the other ranks pass None and just send.  One grouped
    point-to-point exchange (RCCL over xGMI: every rank has its own link to `dst`), nothing padded, no size exchange.
    Returns the concatenation on `dst`, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    gloo = dist.get_backend() == "gloo"
    if gloo and packed.is_cuda:
        packed = packed.cpu()
    if world == 1:
        return packed
    # (both sides through batch_isend_irecv: torch runs batched point-to-point operations on the group's own communicator and plain
    # send / recv on a pair communicator it creates on first use -- a send on the one never meets a receive on the other)
    if rank != dst:
        if packed.numel():
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, packed, dst)]):
                w.wait()
        return None
    totals = [int(x) for x in totals]
    if len(totals) != world or totals[dst] != packed.numel():
        raise ValueError("gather_ragged_known: totals %r do not describe this rank's %d bytes" % (totals, packed.numel()))
    parts = [packed if r == dst else torch.empty(totals[r], dtype=torch.uint8, device=packed.device) for r in range(world)]
    ops = [dist.P2POp(dist.irecv, parts[r], r) for r in range(world) if r != dst and totals[r]]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return torch.cat(parts)


def gather_payloads(dist, kinds, kind_totals, dst=0, joined=None):
    """Several ragged payloads in ONE exchange: `kinds` = this rank's packed uint8 tensors (one per payload kind, in a fixed order),
    `kind_totals[r][k]` = bytes of kind k on rank r as `dst` knows them (None on the other ranks).  Returns, on `dst`, one tensor per
    kind holding that kind of every rank in rank order; None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = joined if joined is not None else (torch.cat(kinds) if len(kinds) > 1 else kinds[0])  # (joined: the kinds are views of one buffer, in order)
    if rank != dst:
        gather_ragged_known(dist, mine, None, dst)
        return None
    if world == 1:
        return list(kinds)
    allb = gather_ragged_known(dist, mine, [sum(int(x) for x in kt) for kt in kind_totals], dst)
    per_kind = [[] for _ in kinds]
    at = 0
    for r in range(world):
        for k in range(len(kinds)):
            n = int(kind_totals[r][k])
            per_kind[k].append(allb[at:at + n])
            at += n
    return [torch.cat(x) for x in per_kind]


class ResultGather:
    """The final gather of a sharded job, both halves: the fixed-size records of every trace to `dst` in ONE collective
    (gather_records), then the variable-length payloads -- traceback strings, rewritten basecalls, decomposition tables -- packed per
    rank (tracyhip_pack_ragged on the device) and shipped in ONE grouped exchange whose sizes `dst` reads off the records it has just
    received (their length columns): no size exchange, nothing padded.

    sizes: traces per rank (shard_range).  ctx: the rank's tracy_amd.Context (packs CUDA tensors; CPU tensors -- the gloo tests --
    are packed with torch).  After gather(): .bytes_last = bytes this rank contributed (records + payloads)."""

    def __init__(self, dist, sizes, ctx=None, dst=0):
        self.dist, self.sizes, self.ctx, self.dst = dist, [int(x) for x in sizes], ctx, dst
        self.scratch = {}
        self.bytes_last = 0

    def _pack(self, key, records, payloads):
        """this rank's payloads packed kind-major into one buffer: (buffer, [bytes per kind])"""
        n, F = int(records.shape[0]), int(records.shape[1])
        flat = records.reshape(-1)
        if payloads and payloads[0][0].is_cuda:
            if self.ctx is None:
                raise RuntimeError("ResultGather: CUDA results need the rank's Context (tracyhip_pack_ragged_multi)")
            need = sum(n * stride * buf.element_size() for buf, stride, _ in payloads)
            out = self.scratch.get(key)
            if out is None or out.numel() < need:
                out = self.scratch[key] = torch.empty(max(need, 1), dtype=torch.uint8, device=payloads[0][0].device)
            return self.ctx.pack_ragged_multi([(buf, stride, flat[col:], F) for buf, stride, col in payloads], n, out=out)
        parts = [pack_ragged(buf, stride, flat[col:][::F][:n])[0] for buf, stride, col in payloads]  # (CPU tensors: the gloo tests)
        return (torch.cat(parts) if len(parts) > 1 else parts[0]), [int(x.numel()) for x in parts]

    def gather(self, records, payloads):
        """records: int32 [n_local, F], contiguous.  payloads: list of (buf, stride_elements, column): trace i uses the first
        records[i, column] elements of buf[i * stride : (i + 1) * stride].  Returns (records of all ranks, [payload bytes of all ranks per
        kind]) on dst, (None, None) elsewhere."""
        dist, dst = self.dist, self.dst
        rank, world = dist.get_rank(), dist.get_world_size()
        assert records.dtype == torch.int32 and records.is_contiguous()
        allrec = gather_records(dist, records, dst=dst, sizes=self.sizes)
        if not payloads:
            self.bytes_last = records.numel() * 4
            return allrec, []
        packed, kind_bytes = self._pack("gather", records, payloads)
        self.bytes_last = records.numel() * 4 + int(packed.numel())
        totals = None
        if rank == dst and world > 1:  # bytes per (rank, kind), read off the gathered length columns: the one host read of the gather
            cols = torch.tensor([c for _, _, c in payloads], device=allrec.device)
            elem = torch.tensor([b.element_size() for b, _, _ in payloads], dtype=torch.int64, device=allrec.device)
            lens = allrec[:, cols].to(torch.int64)
            bounds = [0]
            for x in self.sizes:
                bounds.append(bounds[-1] + x)
            per = torch.stack([lens[bounds[r]:bounds[r + 1]].sum(dim=0) for r in range(world)]) * elem[None, :]
            totals = per.cpu().tolist()
        kinds, at = [], 0
        for nb in kind_bytes:
            kinds.append(packed[at:at + nb])
            at += nb
        got = gather_payloads(dist, kinds, totals, dst, joined=packed)
        return (allrec, got) if rank == dst else (None, None)

    def check_own_block(self, allrec, got, records, payloads):
        """on dst, after gather(): the gathered arrays hold every rank's records, this rank's block bit for bit where shard order puts it, and
        exactly the payload bytes the gathered length columns announce (a cheap in-run check; the other ranks' blocks are compared with
        single-process results in tests/test_shard_gloo.py and tests/test_gpu_bench_ranks.py)"""
        rank = self.dist.get_rank()
        lo = sum(self.sizes[:rank])
        n = int(records.shape[0])
        rec = records.cpu() if (records.is_cuda and not allrec.is_cuda) else records
        ok = int(allrec.shape[0]) == sum(self.sizes) and torch.equal(allrec[lo:lo + n], rec)
        mine, kind_bytes = self._pack("check", records, payloads) if payloads else (None, [])
        at_mine = 0
        for k, (buf, stride, col) in enumerate(payloads):
            elem = buf.element_size()
            lens_all = allrec[:, col].to(torch.int64)
            ok = ok and int(got[k].numel()) == int(lens_all.sum().item()) * elem
            at = int(lens_all[:lo].sum().item()) * elem
            part = mine[at_mine:at_mine + kind_bytes[k]]
            at_mine += kind_bytes[k]
            part = part.cpu() if (part.is_cuda and not got[k].is_cuda) else part
            ok = ok and torch.equal(got[k][at:at + kind_bytes[k]], part)
        return bool(ok)


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
