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


class _Pending:
    """An all_gather in flight: keeps its buffers alive; wait() returns the gathered rows."""

    def __init__(self, work, out, total, keep):
        self.work, self.out, self.total, self.keep = work, out, total, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.out[:self.total]


def gather_rows_async(local, total):
    """gather_rows without waiting: ONE all_gather_into_tensor issued with async_op (RCCL runs it on its own stream,
    ordered behind the work already queued on the current stream), so that the next step's search overlaps it.  The
    caller must not overwrite `local` before wait()."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return _Pending(None, local, total, None)
    world = dist.get_world_size()
    per = (total + world - 1) // world
    pad = local
    if local.numel() < per:
        pad = torch.cat([local, torch.full((per - local.numel(),), -1, dtype=local.dtype, device=local.device)])
    out = torch.empty(per * world, dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, pad, async_op=True)
    return _Pending(work, out, total, pad)


def gather_paths(d_len, d_off, d_child, used, per):
    """all_gather of shortestpath results in TWO collectives: a header block per rank — hop counts and list offsets of
    its `per` rows (the last shard padded) and its payload size — then the packed [v,e,v,...] payloads, padded to the
    largest (read from the gathered headers: 8 bytes per rank come back to the host).  Returns (lengths, offsets into
    the concatenated payload, payload); offsets of rank r are shifted by the payload sizes of the ranks before it."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return d_len, d_off, d_child[:used]
    world = dist.get_world_size()
    dev = d_len.device
    head = torch.empty(2 * per + 1, dtype=torch.int64, device=dev)
    head[:per] = -1
    head[per:2 * per] = 0
    head[:d_len.numel()] = d_len
    head[per:per + d_off.numel()] = d_off
    head[2 * per] = used
    heads = torch.empty(world * (2 * per + 1), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(heads, head)
    heads = heads.view(world, 2 * per + 1)
    sizes = heads[:, 2 * per].tolist()  # the one host read of the gather
    cap = max(max(sizes), 1)
    block = torch.zeros(cap, dtype=d_child.dtype, device=dev)
    block[:used] = d_child[:used]
    blocks = torch.empty(world * cap, dtype=d_child.dtype, device=dev)
    dist.all_gather_into_tensor(blocks, block)
    blocks = blocks.view(world, cap)
    base = torch.zeros(world, dtype=torch.int64)
    base[1:] = torch.cumsum(torch.tensor(sizes[:-1], dtype=torch.int64), 0)
    offs = heads[:, per:2 * per] + base.to(dev)[:, None]
    return heads[:, :per].reshape(-1), offs.reshape(-1), torch.cat([blocks[r, :sizes[r]] for r in range(world)])


def shard_bounds_grouped(total, world, rank, group):
    """shard_bounds for rows that come in groups of `group` consecutive rows (a cross product emitted by a nested-loop
    join: every source's rows in one stretch, match.cpp:467-495): the cut points are moved down to group boundaries, so
    that every rank keeps WHOLE sources — one two-hop ball (or one lane) per source is then built on exactly one GPU."""
    group = max(1, int(group))
    lo, hi = shard_bounds(total, world, rank)
    lo = (lo // group) * group
    hi = total if rank == world - 1 or hi >= total else (hi // group) * group
    return min(lo, total), max(min(hi, total), min(lo, total))
