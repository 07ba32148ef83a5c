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
    pend = sharding.gather_rows_async(torch.from_numpy(ln), total)  # what bench.py overlaps with the next step
    assert torch.equal(pend.wait(), allr)
    # path lists: packed per-rank payloads of different sizes, offsets relative to the rank's own buffer
    paths = ora.lean_shortestpath(V, pairs[lo:hi, 0], pairs[lo:hi, 1])
    offs, child = [], []
    for pth in paths:
        offs.append(len(child))
        child.extend(pth or [])
    per = (total + world - 1) // world
    g_len, g_off, g_child = sharding.gather_paths(torch.from_numpy(ln), torch.tensor(offs, dtype=torch.int64),
                                                  torch.tensor(child + [0] * 7, dtype=torch.int64), len(child), per)
    if rank == 0:
        ln1, ok1 = ora.lean_iterativelength(V, pairs[:, 0], pairs[:, 1])
        ln1[~ok1] = -1
        want = ora.lean_shortestpath(V, pairs[:, 0], pairs[:, 1])
        got, k = [], 0
        for r in range(world):  # gathered blocks are `per` rows each (the last one padded)
            rlo, rhi = sharding.shard_bounds(total, world, r)
            for j in range(rhi - rlo):
                i = r * per + j
                length = int(g_len[i])
                got.append(None if length < 0 else g_child[int(g_off[i]):int(g_off[i]) + 2 * length + 1].tolist())
        q.put((allr.numpy().tolist() == ln1.tolist() and got == want, int(allr.numel())))
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


def test_bare_bench_command_starts_n_ranks():
    """`python bench.py --gpus 2` without a launcher around it must start 2 ranks by itself (round 4: it silently ran
    one and reported n_gpus 1).  --launch-check stops after the ranks have met and counted themselves: no GPU work."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--backend", "gloo", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0's line only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["gpus_arg"] == 2
    # under a launcher (WORLD_SIZE set) nothing is re-launched
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--launch-check"],
                       capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def _worker8(rank, world, port, total, q):
    """One of 8 gloo ranks: its shard's results are a pure function of the global row index, so rank 0 can check the
    gathered arrays without any search."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_bounds(total, world, rank)
    idx = torch.arange(lo, hi, dtype=torch.int64)
    ln = idx % 6 - 1  # -1 = NULL row, else a hop count 0..4
    pend = sharding.gather_rows_async(ln.clone(), total)  # what bench.py issues per step (async, overlapped)
    allr = pend.wait()
    ok = torch.equal(allr, torch.arange(total, dtype=torch.int64) % 6 - 1)
    # ragged path lists: 2 * len + 1 elements per answered row, element j of row i = i * 10 + j
    sizes = torch.where(ln >= 0, 2 * ln + 1, torch.zeros_like(ln))
    offs = torch.cumsum(sizes, 0) - sizes
    used = int(sizes.sum())
    child = torch.zeros(used + 5, dtype=torch.int64)
    for i, (l, o) in enumerate(zip(ln.tolist(), offs.tolist())):
        if l >= 0:
            child[o:o + 2 * l + 1] = (lo + i) * 10 + torch.arange(2 * l + 1)
    per = (total + world - 1) // world
    g_len, g_off, g_child = sharding.gather_paths(ln, offs, child, used, per)
    if rank == 0:
        for r in range(world):
            rlo, rhi = sharding.shard_bounds(total, world, r)
            rows = list(range(rlo, rhi, max(1, (rhi - rlo) // 97))) + [rhi - 1] if rhi > rlo else []
            for i in rows:  # ~100 rows of every rank's block, its first and last among them
                k = r * per + (i - rlo)
                l = int(g_len[k])
                ok = ok and l == i % 6 - 1
                if l >= 0:
                    o = int(g_off[k])
                    ok = ok and g_child[o:o + 2 * l + 1].tolist() == [i * 10 + j for j in range(2 * l + 1)]
        q.put((bool(ok), int(allr.numel())))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_uneven_shards_gathers():
    """world_size 8 over gloo, 65,536 rows (even shards) and 65,535 (the last shard one row short: padded blocks): the
    async length gather bench.py overlaps with the next step, and the two-collective gather of ragged path lists."""
    for total in (65536, 65535):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker8, args=(r, 8, port, total, q)) for r in range(8)]
        for p in procs:
            p.start()
        same, n = q.get(timeout=300)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert same and n == total


def test_grouped_shards_keep_whole_sources():
    # a cross product of S sources x D rows each over N ranks: cut points on source boundaries, everything covered once
    for sources, per_source, world in ((2048, 1024, 8), (2048, 1024, 3), (7, 100, 4), (5, 1, 8), (1, 10, 2)):
        total = sources * per_source
        cover = []
        for r in range(world):
            lo, hi = sharding.shard_bounds_grouped(total, world, r, per_source)
            assert lo % per_source == 0 and (hi % per_source == 0 or hi == total) and lo <= hi
            cover.extend(range(lo, hi))
        assert cover == list(range(total))
