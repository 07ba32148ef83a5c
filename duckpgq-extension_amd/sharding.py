"""Multi-GPU plumbing of the path (one process per GPU, torch.distributed over RCCL; gloo in the CPU tests).

The path shards by independent units: every (src,dst) pair is a pure function of (CSR, src, dst)
(SURVEY.md §8e).  So: CSR replicated (one broadcast per query), pairs cut into contiguous per-rank ranges, no
collective inside the search, one all_gather of the per-pair results at the end.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, world, rank):
    """Contiguous range [lo, hi) of `total` rows owned by `rank` (ceil split like SURVEY §8e)."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def broadcast_csr(arrays, device, src=0):
    """arrays: dict name -> int64 torch tensor on rank `src` (None elsewhere).  Returns the same dict on every rank,
    resident on `device`.  Two collectives: sizes, then payloads."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    names = sorted(arrays) if rank == src else None
    if world > 1:
        box = [names]
        dist.broadcast_object_list(box, src)
        names = box[0]
    sizes = torch.zeros(len(names), dtype=torch.int64, device=device)
    if rank == src:
        sizes[:] = torch.tensor([arrays[n].numel() for n in names], dtype=torch.int64)
    if world > 1:
        dist.broadcast(sizes, src)
    out = {}
    for n, sz in zip(names, sizes.tolist()):
        t = arrays[n].to(device) if rank == src else torch.empty(sz, dtype=torch.int64, device=device)
        if world > 1:
            dist.broadcast(t, src)
        out[n] = t
    return out


def gather_rows(local, total):
    """all_gather of equally sized per-rank result blocks (the last shard is padded), trimmed to `total` rows."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    per = (total + world - 1) // world
    pad = local
    if local.numel() < per:
        pad = torch.cat([local, torch.full((per - local.numel(),), -1, dtype=local.dtype, device=local.device)])
    parts = [torch.empty(per, dtype=local.dtype, device=local.device) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat(parts)[:total]


def gather_paths(d_len, d_off, d_child, used, per):
    """all_gather of shortestpath results: per-pair hop counts and list offsets (equal blocks of `per` rows, the last
    shard padded) and the packed [v,e,v,...] payloads (ragged: sizes first, then blocks padded to the largest).
    Returns (lengths, offsets into the concatenated payload, payload); offsets of rank r are shifted by the payload
    sizes of the ranks before it."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return d_len, d_off, d_child[:used]
    world = dist.get_world_size()
    dev = d_len.device

    def pad_to(t, k, fill):
        return t if t.numel() >= k else torch.cat([t, torch.full((k - t.numel(),), fill, dtype=t.dtype, device=dev)])

    lens = [torch.empty(per, dtype=d_len.dtype, device=dev) for _ in range(world)]
    offs = [torch.empty(per, dtype=d_off.dtype, device=dev) for _ in range(world)]
    dist.all_gather(lens, pad_to(d_len, per, -1))
    dist.all_gather(offs, pad_to(d_off, per, 0))
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([used], dtype=torch.int64, device=dev))
    sizes = [int(x.item()) for x in sizes]
    cap = max(max(sizes), 1)
    blocks = [torch.empty(cap, dtype=d_child.dtype, device=dev) for _ in range(world)]
    dist.all_gather(blocks, pad_to(d_child[:used], cap, 0))
    base, shifted = 0, []
    for r in range(world):
        shifted.append(offs[r] + base)
        base += sizes[r]
    return torch.cat(lens), torch.cat(shifted), torch.cat([b[:k] for b, k in zip(blocks, sizes)])
