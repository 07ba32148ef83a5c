"""N>1 path on CPU: world_size 2 over gloo.  The GPU search itself cannot run here, so each rank answers its shard
with the CPU oracle (test infrastructure) — what is under test is the distributed plumbing bench.py uses:
CSR broadcast, contiguous pair sharding, and the final all_gather of per-pair lengths."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from duckpgq_extension_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pgq_oracle import OracleCSR
    dev = torch.device("cpu")
    arrays = None
    V = 400
    if rank == 0:
        rng = np.random.default_rng(1)
        s, d = rng.integers(0, V, 2500), rng.integers(0, V, 2500)
        order = np.argsort(s, kind="stable")
        off = np.zeros(V + 1, dtype=np.int64)
        np.cumsum(np.bincount(s, minlength=V), out=off[1:])
        arrays = {"off": torch.from_numpy(off), "adj": torch.from_numpy(d[order].astype(np.int64))}
    arrays = sharding.broadcast_csr(arrays, dev)  # CSR replicated on every rank
    pairs = np.random.default_rng(4).integers(0, V, (total, 2))  # same global list everywhere
    lo, hi = sharding.shard_bounds(total, world, rank)
    ora = OracleCSR.adopt(V, arrays["off"].numpy(), arrays["adj"].numpy())
    ln, ok = ora.lean_iterativelength(V, pairs[lo:hi, 0], pairs[lo:hi, 1])
    ln[~ok] = -1
    allr = sharding.gather_rows(torch.from_numpy(ln), total)
    if rank == 0:
        ln1, ok1 = ora.lean_iterativelength(V, pairs[:, 0], pairs[:, 1])
        ln1[~ok1] = -1
        q.put((allr.numpy().tolist() == ln1.tolist(), int(allr.numel())))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_pairs_gather_equals_single_process():
    for total in (1000, 1001):  # even and ragged split
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
        for p in procs:
            p.start()
        same, n = q.get(timeout=120)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert same and n == total


def test_shard_bounds_cover_everything():
    for total in (0, 1, 7, 8192, 65536, 65537):
        for world in (1, 2, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = sharding.shard_bounds(total, world, r)
                assert 0 <= lo <= hi <= total
                cover.extend(range(lo, hi))
            assert cover == list(range(total))
