#!/usr/bin/env python3
"""bench.py — DuckPGQ's path-finding hot path on MI355X (contract: one JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over this rank's batch of (src,dst) pairs, inputs resident in HBM:
pgq_iterativelength_bulk_device (pair-centric pre-pass for scattered pairs, lane-batched MS-BFS for cross products and
for whatever the pre-pass leaves open), followed for N > 1 by the RCCL all_gather of the per-pair results (the only
inter-GPU traffic; the CSR is replicated).

Default workload = BASELINE.json configs[3] (the config the metric "MS-BFS MTEPS + src-dst pairs/sec, SNB SF100,
1/2/4/8 GPU" is quoted on; it fits one GPU): synthetic LDBC-SNB-SF100-shaped Person-knows-Person graph (V=448,626,
39.88 M symmetric CSR entries), iterativelength on 65,536 random pairs (`default_rng(4)`).
  N = 1:  the top-level fields describe that workload (= leg "prepass": every row is answered by the pair-centric
          kernels); `legs` holds it next to one leg per other BASELINE config and call shape (`legs_by_config` maps them):
            msbfs_cross       the same graph in the binder's call shape (match.cpp:467-495: a cross product of endpoints —
                              2048 distinct sources x 1024 destinations = 2.1 M rows, grouped by source), as the library routes
                              it (round 6: the source-centric kernel k_src_ball, one two-hop ball per source run)
            msbfs_cross_lanes the same rows forced through the lane-batched MS-BFS (handle option ball = 0): the frontier
                              expansion kernels' own figures (rounds 1-5's msbfs_cross)
            msbfs_cross_shuffled  the same rows in random order (a hash join's output): sorted by source for k_src_ball
            msbfs_cross_rmat22  the same call shape on R-MAT-22 (configs[1]'s graph): 2048 x 1024 rows, frontier arrays far past
                              the 256-MiB Infinity Cache — the MS-BFS level kernels against DRAM
            snb_paths         configs[2]: shortestpath + reconstruction, 4096 pairs, same CSR
            rmat22            configs[1]: R-MAT scale 22, iterativelength, 1024 pairs (own graph: ~15 s of generation)
            forest_cheapest   configs[4]: weighted cheapest path on a 2^24-vertex reply forest, 4096 pairs
            cheapest_general  cheapest_path_length of 4096 pairs on the knows graph with int64 weights (one step)
          each with its own ms/step, pairs/s, MTEPS, roofline and a bounded CPU comparison of the TIMED output
          (--config-legs '' / --no-legs / --cheapest-pairs 0 drop them; the whole default command takes ~2-3 minutes).
          `first_call_ms` (per leg): a FRESH CSR handle (pgq_csr_upload_device) -> its first search, no warm-up call, median
          over 5 handles — the reference's CSR lives for one query (iterative_length_function_data.cpp:27), so this is what a
          query sees; `ms_per_step` is the steady state of a handle that has answered the same call before.
          `legs_summary` (LAST key of the line, and mirrored as top-level scalars leg_<name>_ms / _frac / _first_ms):
          {leg: [ms_per_step, chain frac, whole-step frac, rows compared with the CPU port, rows equal, first_call_ms]}.
  N > 1:  --scaling strong by default (configs[3] is 65,536 pairs in total, cut across the GPUs); the weak figure
          (65,536 pairs on every GPU) is measured in the same run and reported under "weak".
Other workloads as the main line: --workload rmat22 | snb_paths | forest_cheapest (--scale 28 = configs[4]'s named size) |
snb_cross | snb_cross_allv | snb_cheapest.

value    = (src, dst) pairs answered per second, whole job (metric "src_dst_pairs_per_s"; BASELINE's metric is "MS-BFS
           MTEPS + src-dst pairs/sec").  Beside it: `mteps_physical` = adjacency entries the kernels really scanned per
           second / 1e6 (a hardware rate), and `mteps_logical` = traversed edges / second / 1e6 where the traversed edges
           of a pair = out-degrees of all vertices its own level-synchronous BFS expands up to the level that reaches
           dst (all levels if unreachable) — what the reference's per-pair lane traverses; a pure function of (graph,
           src, dst), counted once on the GPU outside the timed region (pgq_traversed_edges_bulk_device) and pinned
           against the CPU oracle in tests/.  The logical figure counts work the pair-centric kernels AVOID: it is not
           a hardware throughput and is not the headline (it was round 3's `value`; kept as `msbfs_mteps`).
roofline = the leg's whole LAUNCH CHAIN (schema 5): the algorithmic bytes of all its kernel classes (DESIGN.md has the
           formulas) over the sum of their launch durations, measured in an untimed pass with one batch in flight and HIP
           events around every launch on the library's own stream; `dominant_kernel` = the single class with the most
           time (round 4's top level), `step` = the same bytes over the wall time of the timed region, `traffic` = HBM
           bytes per step of that chain from the committed PMC passes (profiles/pmc_<workload>.json).
cpu_baseline = the literal restatement of the reference UDF (oracle/, 512-lane bitsets, 2048-row chunks) timed on this
           box's host cores on a bounded sample of the same pairs, same MTEPS definition, compared row by row with
           the output of the TIMED steps (the output buffer is poisoned before the timed loop).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); the measured copy ceiling is reported beside it
POISON = -7             # written into the result buffers before the timed loop: a step that writes nothing is caught

OPS = {"snb_sf100": "iterativelength", "rmat22": "iterativelength", "rmat22_cross": "iterativelength", "snb_paths": "shortestpath+reconstruction",
       "forest_cheapest": "cheapest_path_length", "snb_cheapest": "cheapest_path_length",
       "snb_cross": "iterativelength", "snb_cross_allv": "iterativelength"}
DEFAULT_PAIRS = {"snb_sf100": 65536, "rmat22": 1024, "rmat22_cross": 2048 * 1024, "snb_paths": 4096, "forest_cheapest": 4096, "snb_cheapest": 4096,
                 "snb_cross": 2048 * 1024, "snb_cross_allv": 0}
PAIR_SEED = {"snb_sf100": 4, "rmat22": 2, "rmat22_cross": 29, "snb_paths": 3, "forest_cheapest": 5, "snb_cheapest": 6, "snb_cross": 7,
             "snb_cross_allv": 8}
CHEAPEST = ("forest_cheapest", "snb_cheapest")  # weighted workloads: value = pairs/s
SNB = ("snb_sf100", "snb_paths", "snb_cheapest", "snb_cross", "snb_cross_allv")
EXPANSION = ("push", "pull", "pull_hub", "pull_sparse")  # the MS-BFS frontier-expansion kernel classes
KERNEL_OF = {"meet": "k_meet3", "meet4": "k_meet4d", "bibfs": "k_bibfs", "pull_sparse": "k_pull_lanes", "ball": "k_src_ball"}
PREPASS = ("ball", "meet", "meet4", "bibfs")  # the source-centric kernel and the pair-centric ones: k_src_ball, k_meet3, k_meet4d / k_meet4, k_bibfs


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="snb_sf100", choices=sorted(OPS))
    ap.add_argument("--pairs-per-gpu", type=int, default=0, help="0 = the BASELINE config's pair count")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: pairs-per-gpu rows on every rank; strong: pairs-per-gpu rows in total, cut across ranks "
                         "(default: strong when more than one GPU runs — configs[3] is 65,536 pairs in total)")
    ap.add_argument("--scale", type=int, default=0, help="override graph scale (rmat scale / forest log2 V); tests")
    ap.add_argument("--snb-vertices", type=int, default=448626)
    ap.add_argument("--snb-friendships", type=int, default=19_940_000)
    ap.add_argument("--cross-sources", type=int, default=2048, help="snb_cross / msbfs_cross leg: distinct sources")
    ap.add_argument("--cross-dests", type=int, default=1024, help="snb_cross / msbfs_cross leg: destinations per source")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE", help="pgq_set_option before the run (experiments; recorded in config.options)")
    ap.add_argument("--trace-steps", action="store_true", help="debugging: print every timed step's end time and counters to stderr")
    ap.add_argument("--cross-shuffle", action="store_true", help="snb_cross / rmat22_cross: rows in random order (a hash join's output) instead of grouped by source")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="N = 1 default workload: skip the msbfs_cross and cheapest_general legs")
    ap.add_argument("--cheapest-pairs", type=int, default=4096, help="cheapest_general leg: pairs (0: skip the leg)")
    ap.add_argument("--config-legs", default="snb_paths,rmat22,forest_cheapest",
                    help="N = 1 default workload: the other BASELINE configs run as legs of the same line (comma list; '' = none)")
    ap.add_argument("--leg-rmat-scale", type=int, default=22, help="rmat22 leg: R-MAT scale (tests shrink it)")
    ap.add_argument("--no-first-call", action="store_true", help="skip the first-call-on-a-fresh-handle measurements")
    ap.add_argument("--first-call-handles", type=int, default=5, help="fresh CSR handles per first_call_ms figure")
    ap.add_argument("--leg-forest-scale", type=int, default=24, help="forest_cheapest leg: log2 V (tests shrink it)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs timed on one CPU thread (0 = 8192 snb / 1024 rmat)")
    ap.add_argument("--weights", default="int64", choices=["int64", "double"], help="forest_cheapest / snb_cheapest: weight type")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, count them with one all_reduce and print {n_gpus}: the CPU test of the "
                         "launcher (no GPU work)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (one process per GPU,
    torch.distributed.run on a free loopback port) with the same arguments, and hand their output through — rank 0's
    JSON line is the only line any rank prints.  Under `python -m torch.distributed.run ... bench.py --gpus N` WORLD_SIZE
    is set and this is not entered."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def launch_check(a):
    """The ranks meet, count themselves and rank 0 prints the count: everything of an N-rank run but the GPU work."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    n = torch.ones(1, dtype=torch.int64)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist.all_reduce(n)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": int(n[0]), "world_size": world, "gpus_arg": a.gpus}), flush=True)


def build_graph(a):
    """(name, V, offsets, adj, edge ids, weights or None, seconds) of a.workload's graph (numpy, seeded)."""
    from duckpgq_extension_amd import graphgen
    t0 = time.time()
    w = None
    if a.workload in SNB:
        V, s, d = graphgen.snb_knows_like(a.snb_vertices, a.snb_friendships, seed=100)
        name = "snb_sf100_knows(V=%d)" % V
        if a.workload == "snb_cheapest":  # the general-graph case of cheapest_path_length: weights 1..999 on the knows graph
            w = np.random.default_rng(6).integers(1, 1000, len(s))
            if a.weights == "double":
                w = w.astype(np.float64) / 7.0
            name += ",%s w" % a.weights
    elif a.workload in ("rmat22", "rmat22_cross"):
        V, s, d = graphgen.rmat(a.scale or 22, seed=22)
        name = "rmat%d_ef16" % (a.scale or 22)
    else:
        V, s, d = graphgen.reply_forest(1 << (a.scale or 24), seed=5)
        w = np.random.default_rng(5).integers(1, 1000, len(s))
        if a.weights == "double":
            w = w.astype(np.float64) / 7.0
        name = "reply_forest(V=2^%d,%s w)" % (a.scale or 24, a.weights)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    if w is not None:
        w = w[eid]
    return name, V, off, adj, eid, w, time.time() - t0


def cross_pairs(V, total, sources, seed):
    """The binder's call shape (match.cpp:467-495): a cross product of endpoints.  `sources` distinct sources, each
    with total / sources random destinations; rows grouped by source like a nested-loop join emits them."""
    rng = np.random.default_rng(seed)
    src = rng.choice(V, size=sources, replace=False)
    per = max(1, total // sources)
    s = np.repeat(src, per)
    d = rng.integers(0, V, len(s))
    return np.stack([s, d], axis=1).astype(np.int64)


def make_pairs(a, V, total, off, adj):
    """One global, seeded pair list.  The reply forest gets destinations that are ancestors of their sources (uniform
    pairs are almost never connected there: the search would only measure the dead-end shortcut)."""
    rng = np.random.default_rng(PAIR_SEED[a.workload])
    if a.workload == "rmat22_cross":  # the leg msbfs_cross_rmat22 as a workload of its own (same rows: seed 7 + 22)
        return cross_pairs(V, total, a.cross_sources * max(1, total // (a.cross_sources * a.cross_dests)), PAIR_SEED["snb_cross"] + 22)
    if a.workload == "snb_cross":
        cp = cross_pairs(V, total, a.cross_sources * max(1, total // (a.cross_sources * a.cross_dests)), PAIR_SEED[a.workload])
        return cp[rng.permutation(len(cp))] if a.cross_shuffle else cp
    if a.workload == "snb_cross_allv":  # 32 sources x every vertex as destination
        src = rng.choice(V, size=32, replace=False)
        return np.stack([np.repeat(src, V), np.tile(np.arange(V, dtype=np.int64), 32)], axis=1).astype(np.int64)
    if a.workload != "forest_cheapest":
        return rng.integers(0, V, size=(total, 2))
    deg = np.diff(off)
    cand = np.nonzero(deg > 0)[0]  # non-roots: exactly one out-edge (child -> parent)
    src = cand[rng.integers(0, len(cand), total)]
    dst = src.copy()
    hops = rng.integers(1, 9, total)
    for h in range(8):
        move = (hops > h) & (deg[dst] > 0)
        dst[move] = adj[off[dst[move]]]
    # one pair in eight keeps a uniform (mostly unreachable) destination
    miss = rng.random(total) < 0.125
    dst[miss] = rng.integers(0, V, int(miss.sum()))
    return np.stack([src, dst], axis=1)


class Bench:
    """One rank's state: the replicated CSR and the timing helpers."""

    def __init__(self, a):
        import torch
        import torch.distributed as dist
        import duckpgq_extension_amd as pgq
        from duckpgq_extension_amd import sharding
        self.a, self.torch, self.dist, self.pgq, self.sharding = a, torch, dist, pgq, sharding
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=a.backend, rank=self.rank, world_size=self.world)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(local % ndev)
        self.dev = torch.device("cuda", local % ndev)
        pgq.load_hip().pgq_init(local % ndev)

    def sync_all(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, wl, csr, mine_t, total_pairs, steps, warmup, paths=False, cheapest=False):
        """Times `steps` passes over this rank's rows `mine_t` ([n, 2] device tensor) and returns the measurements:
        wall time of the timed region, statistics of the timed region and of an isolated profiled pass, the output of
        the timed steps and the traversed-edge counts."""
        torch, dist, pgq, sharding = self.torch, self.dist, self.pgq, self.sharding
        world, dev = self.world, self.dev
        n = mine_t.shape[0]
        d_src, d_dst = mine_t[:, 0].contiguous(), mine_t[:, 1].contiguous()
        # two result buffers: the gather of step k (async, on RCCL's stream) overlaps the search of step k + 1
        d_len = [torch.empty(n, dtype=torch.int64, device=dev) for _ in range(2)]
        d_te = torch.zeros(n, dtype=torch.int64, device=dev)
        child_cap = n * 64
        d_off = d_child = d_val = d_ok = None
        if paths:
            d_off = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(2)]
            d_child = [torch.empty(child_cap, dtype=torch.int64, device=dev) for _ in range(2)]
        if cheapest:
            d_val = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(2)]
            d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)
        pending = [None, None]
        per = (total_pairs + world - 1) // world

        def step(k):
            b = k & 1
            if pending[b] is not None:  # the gather that last read this buffer
                pending[b].wait()
                pending[b] = None
            if paths:
                rc, used = csr.shortestpath_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len[b].data_ptr(),
                                                     d_off[b].data_ptr(), d_child[b].data_ptr(), child_cap)
                assert rc == 0, pgq.load_hip().pgq_last_error()
                if world > 1:
                    sharding.gather_paths(d_len[b], d_off[b], d_child[b], used, per)
            elif cheapest:
                csr.cheapest_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_val[b].data_ptr(), d_ok.data_ptr())
                if world > 1:
                    pending[b] = sharding.gather_rows_async(d_val[b], total_pairs)
            else:
                csr.iterativelength_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len[b].data_ptr())
                if world > 1:  # final RCCL gather of the per-pair results (xGMI)
                    pending[b] = sharding.gather_rows_async(d_len[b], total_pairs)

        def drain():
            for b in (0, 1):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None

        # ---- work units (outside the timed region) ----
        ref_len = None
        if not cheapest:
            csr.traversed_edges_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len[0].data_ptr(), d_te.data_ptr())
            ref_len = d_len[0].clone()
        te_local = int(d_te.sum().item())
        for k in range(warmup):
            step(k)
        drain()
        pgq.set_option("profile", 0)
        for b in (0, 1):  # poison: the comparison below only passes if the TIMED steps wrote every row
            d_len[b].fill_(POISON)
            if cheapest:
                d_val[b].fill_(POISON)
        pgq.reset_stats()
        self.sync_all()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
            if self.a.trace_steps:  # debugging: every step on its own (the figure below then includes a sync per step)
                self.sync_all()
                print("[bench] %s step %d done at +%.3f ms, stats %s" % (wl, k, (time.perf_counter() - t0) * 1e3, {
                    kk: vv for kk, vv in pgq.get_stats().items() if kk in ("ball_calls", "meet_pairs", "levels", "batches", "host_waits")}), file=sys.stderr)
        drain()
        self.sync_all()
        elapsed = time.perf_counter() - t0
        stats = pgq.get_stats()
        last = (steps - 1) & 1
        out_len = d_len[last].clone()
        if cheapest:
            reach = int(d_ok.sum().item())
            out_val = d_val[last].clone()
        else:
            assert bool((out_len == ref_len).all()), "timed output differs from the traversed-edge accounting pass"
            if steps > 1:
                assert bool((d_len[last ^ 1] == ref_len).all()), "timed output differs from the accounting pass"
            reach = int((out_len >= 0).sum().item())
            out_val = None
        # Untimed pass with one batch in flight and HIP events around every launch (recorded on the library's own stream):
        # nothing overlaps, so a launch's event duration is that kernel's duration.
        n_streams, n_relax_streams = int(pgq.get_option("streams")), int(pgq.get_option("relax_streams"))
        pgq.set_option("streams", 1)
        pgq.set_option("relax_streams", 1)
        pgq.set_option("profile", 1)
        iso_steps = max(1, min(steps, 3))
        step(0)
        drain()
        pgq.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(iso_steps):
            step(k)
        drain()
        torch.cuda.synchronize()
        iso_elapsed = time.perf_counter() - t0
        iso = pgq.get_stats()
        pgq.set_option("profile", 0)
        pgq.set_option("streams", n_streams)
        pgq.set_option("relax_streams", n_relax_streams)
        res = {"n": n, "elapsed": elapsed, "stats": stats, "iso": iso, "iso_steps": iso_steps, "iso_elapsed": iso_elapsed,
               "te_local": te_local, "reach": reach, "out_len": out_len, "out_val": out_val, "d_ok": d_ok, "d_te": d_te,
               "steps": steps}
        if paths:  # what the isolated pass (same rows, same answers) left in buffer 0: offsets + payload of every list
            res["out_off"], res["out_child"] = d_off[(iso_steps - 1) & 1].clone(), d_child[(iso_steps - 1) & 1].clone()
            assert bool((d_len[(iso_steps - 1) & 1] == out_len).all())
        return res

    def first_call(self, make_csr, mine_t, handles, paths=False, cheapest=False):
        """ms from `a fresh handle exists` to `its first search has returned`, median over `handles` fresh handles: no warm-up
        call, no route memo, no level plan, calibration included.  The upload itself is timed separately (`upload_ms`)."""
        torch = self.torch
        n = mine_t.shape[0]
        d_src, d_dst = mine_t[:, 0].contiguous(), mine_t[:, 1].contiguous()
        d_len = torch.empty(n, dtype=torch.int64, device=self.dev)
        d_off = torch.zeros(n, dtype=torch.int64, device=self.dev) if paths else None
        d_child = torch.empty(n * 64, dtype=torch.int64, device=self.dev) if paths else None
        d_val = torch.zeros(n, dtype=torch.int64, device=self.dev) if cheapest else None
        d_ok = torch.zeros(n, dtype=torch.uint8, device=self.dev) if cheapest else None
        first, up = [], []
        for _ in range(max(1, handles)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            csr = make_csr()
            t1 = time.perf_counter()
            if paths:
                csr.shortestpath_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr(), d_off.data_ptr(), d_child.data_ptr(), n * 64)
            elif cheapest:
                csr.cheapest_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_val.data_ptr(), d_ok.data_ptr())
            else:
                csr.iterativelength_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr())
            t2 = time.perf_counter()
            first.append((t2 - t1) * 1e3)
            up.append((t1 - t0) * 1e3)
            del csr
        return {"first_call_ms": float(np.median(first)), "first_call_ms_first_handle": round(first[0], 4), "first_call_ms_all": [round(x, 4) for x in first],
                "upload_ms": float(np.median(up)), "handles": len(first),
                "what": "fresh pgq_csr_upload_device handle -> its first search call returned, no warm-up call; median over the handles. The first "
                        "handle of a graph shape in the process calibrates (first_call_ms_first_handle); the later ones find what it "
                        "measured in the per-shape cache (option calibration_cache), like a second query over the same tables"}

    def reduce(self, m):
        """max over ranks of the timed region, sums of the work units."""
        torch, dist = self.torch, self.dist
        el = torch.tensor([m["elapsed"]], dtype=torch.float64, device=self.dev)
        tot = torch.tensor([float(m["te_local"]), float(m["stats"]["edges_scanned"]), float(m["reach"])],
                           dtype=torch.float64, device=self.dev)
        if self.world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        return float(el[0]), float(tot[0]), float(tot[1]), int(tot[2])


def roofline_of(m, workload, copy_gbps, elapsed):
    """Top level = the leg's whole LAUNCH CHAIN: the algorithmic bytes of all its kernel classes over the sum of their
    launch durations (isolated pass: one batch in flight, HIP events around every launch) — for the default workload that
    is k_meet3 + the bit-map kernel (+ k_bibfs), the figure the round-4 review asked to see first; `dominant_kernel` is the
    single class with the most time (round 4's top level), `step` the same bytes over the wall time of the timed region."""
    iso, steps, iso_steps = m["iso"], m["steps"], m["iso_steps"]
    kms, kb, kl = iso["kernel_ms"], iso["algo_bytes"], iso["launches"]
    dom = max(kms, key=lambda k: kms[k])
    ach = kb[dom] / 1e9 / (kms[dom] / 1e3) if kms[dom] > 0 else 0.0
    pmc_file, pmc = os.path.join(ROOT, "profiles", "pmc_%s.json" % workload), {}
    if os.path.exists(pmc_file):  # written by tools/pmc_summary.py from separate rocprofv3 --pmc passes of this command
        try:
            pmc = json.load(open(pmc_file))
        except Exception:
            pmc = {}
    traffic_src = ("profiles/pmc_%s.json (separate rocprofv3 --pmc passes, committed; not collected in this run)" % workload) if pmc else None
    traffic = pmc.get(dom, {}).get("hbm_bytes_per_launch")
    traffic_gbps = float(traffic) / (kms[dom] / max(kl[dom], 1) * 1e-3) / 1e9 if traffic and kms[dom] > 0 else None
    dominant = {"kernel": KERNEL_OF.get(dom, "k_" + dom), "achieved": ach, "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                "traffic_GBps": traffic_gbps, "traffic_frac": (traffic_gbps / HBM_PEAK_GBPS) if traffic_gbps else None,
                "launches": int(kl[dom]), "avg_launch_ms": kms[dom] / max(kl[dom], 1),
                "algorithmic_bytes_per_launch": kb[dom] / max(kl[dom], 1)}
    step_bytes = sum(m["stats"]["algo_bytes"].values())
    prepass = [k for k in PREPASS if kms.get(k, 0.0) > 0]
    chain = prepass if (prepass and dom in PREPASS) else [k for k in kms if kms[k] > 0]
    if chain is prepass and kms.get("prep", 0.0) > 0:  # rows sorted by source in front of the source-centric kernel (ball_sort): part of the chain
        chain = prepass = prepass + ["prep"]
    c_ms, c_b = sum(kms[k] for k in chain), sum(kb[k] for k in chain)
    c_gbps = c_b / 1e9 / (c_ms / 1e3) if c_ms > 0 else 0.0
    no_model = [k for k in chain if kb.get(k, 0.0) <= 0]
    c_traffic = None
    try:  # PMC traffic of the whole chain per step (tools/pmc_summary.py)
        c_traffic = pmc[("prepass_chain" if chain is prepass else "chain")]["hbm_bytes_per_step"]
    except Exception:
        c_traffic = None
    roof = {"bound": "hbm", "kernel": "launch chain: " + " + ".join(KERNEL_OF.get(k, "k_" + k) for k in chain),
            "achieved": c_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": c_gbps / HBM_PEAK_GBPS,
            "traffic": c_traffic, "traffic_source": traffic_src, "per": "step",
            "classes": chain, "ms_per_step": c_ms / iso_steps, "algorithmic_bytes_per_step": c_b / iso_steps,
            "classes_without_byte_model": no_model,
            "measured_copy_GBps": copy_gbps,
            "timing": "HIP events around every launch, one batch in flight, untimed pass; bytes = the kernels' own algorithmic "
                      "model (DESIGN.md 3: for the pair-centric kernels 4 B per adjacency entry walked + 16 B per slot "
                      "descriptor + 64 B per row — NOT SURVEY 8(d)'s per-level formula: that leg runs no BFS level); the SF100 "
                      "adjacency (160 MB padded) and a 2048-lane frontier array (115 MB) are Infinity-Cache-sized: rates "
                      "above the measured copy ceiling are fabric delivery, not DRAM",
            "ms_per_step_of_that_pass": m["iso_elapsed"] / iso_steps * 1e3,
            "dominant_kernel": dominant,
            # all kernel classes' algorithmic bytes over the wall time of the timed region
            "step": {"algorithmic_bytes": step_bytes / steps, "GBps": step_bytes / 1e9 / elapsed,
                     "frac": step_bytes / 1e9 / elapsed / HBM_PEAK_GBPS}}
    if prepass:  # kept under its round-4 name
        p_ms, p_b = sum(kms[k] for k in prepass), sum(kb[k] for k in prepass)
        roof["prepass_chain"] = {"classes": prepass, "ms_per_step": p_ms / iso_steps, "algorithmic_bytes_per_step": p_b / iso_steps,
                                 "GBps": p_b / 1e9 / (p_ms / 1e3), "frac": p_b / 1e9 / (p_ms / 1e3) / HBM_PEAK_GBPS,
                                 "traffic_per_step": pmc.get("prepass_chain", {}).get("hbm_bytes_per_step")}
    exp_ms = sum(kms.get(k, 0.0) for k in EXPANSION)
    exp_b = sum(kb.get(k, 0.0) for k in EXPANSION)
    if exp_ms > 0:  # the MS-BFS frontier-expansion kernels (top-down + bottom-up) together
        roof["frontier_expansion"] = {"classes": [k for k in EXPANSION if kms.get(k, 0.0) > 0],
                                      "ms_per_step": exp_ms / iso_steps, "GBps": exp_b / 1e9 / (exp_ms / 1e3),
                                      "frac": exp_b / 1e9 / (exp_ms / 1e3) / HBM_PEAK_GBPS}
    by_kernel = {k: {"ms_per_step": round(kms[k] / iso_steps, 4),
                     "GBps": round(kb[k] / 1e9 / (kms[k] / 1e3), 1) if kb[k] > 0 else None,
                     "frac": round(kb[k] / 1e9 / (kms[k] / 1e3) / HBM_PEAK_GBPS, 4) if kb[k] > 0 else None,
                     "launches_per_step": kl[k] / iso_steps} for k in kms if kms[k] > 0}
    return roof, by_kernel


def leg_summary(bench, m, workload, total_pairs, copy_gbps):
    elapsed, te_total, scanned, reach = bench.reduce(m)
    steps, stats = m["steps"], m["stats"]
    roof, by_kernel = roofline_of(m, workload, copy_gbps, elapsed)
    return {"ms_per_step": elapsed / steps * 1e3, "pairs_per_s": total_pairs * steps / elapsed,
            "mteps_logical": te_total * steps / elapsed / 1e6, "mteps_physical": scanned / elapsed / 1e6,
            "traversed_edges_per_step": te_total, "physical_edges_scanned_per_step": scanned / steps,
            "reachable_pairs": reach,
            "rows_answered_by_prepass_per_step": stats["meet_pairs"] / max(steps, 1),
            "levels_per_step": stats["levels"] / max(steps, 1),
            "push_pull_levels": [stats["push_levels"] // max(steps, 1), stats["pull_levels"] // max(steps, 1)],
            "deferred_pairs_per_step": stats["deferred_pairs"] / max(steps, 1),
            "roofline": roof, "roofline_by_kernel": by_kernel}, elapsed


def config_leg(bench, a, wl, copy_gbps, snb_graph, snb_csr):
    """One more BASELINE config as a leg of the N = 1 line (round-4 review: C2, C3 and C5 existed only as builder-run
    files): its own graph (rmat22, forest_cheapest) or the default workload's (snb_paths), the config's own pair list, the
    same timed loop / poison / isolated pass, and a bounded CPU comparison of the TIMED output."""
    import copy
    torch, pgq, dev = bench.torch, bench.pgq, bench.dev
    a2 = copy.copy(a)
    a2.workload = wl
    a2.scale = {"rmat22": a.leg_rmat_scale, "forest_cheapest": a.leg_forest_scale}.get(wl, 0)
    a2.pairs_per_gpu = 0
    cheapest, paths = wl in CHEAPEST, wl == "snb_paths"
    if wl in SNB:
        name, V, off, adj, eid, w, gen_s = snb_graph[:7]
        csr, t_keep, upload_s, t_w = snb_csr, None, None, None
    else:
        name, V, off, adj, eid, w, gen_s = build_graph(a2)
        t_keep = [torch.from_numpy(x).to(dev) for x in (off, adj, eid)]
        t_w = torch.from_numpy(np.ascontiguousarray(w).view(np.int64)).to(dev) if w is not None else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        csr = pgq.DeviceCSR.from_device_ptrs(V, t_keep[0].data_ptr(), t_keep[1].data_ptr(), t_keep[2].data_ptr(),
                                             t_w.data_ptr() if t_w is not None else 0,
                                             (2 if a2.weights == "double" else 1) if t_w is not None else 0)
        upload_s = time.perf_counter() - t0
    total = DEFAULT_PAIRS[wl]
    pr = np.ascontiguousarray(make_pairs(a2, V, total, off, adj)).astype(np.int64)
    steps, warmup = max(2, min(a.steps, 10)), min(a.warmup, 2)
    pr_t = torch.from_numpy(pr).to(dev)
    make_csr = None
    if wl in SNB:
        t_off, t_adj, t_eid = snb_graph[-1]  # the default workload's device arrays
        make_csr = lambda: pgq.DeviceCSR.from_device_ptrs(V, t_off.data_ptr(), t_adj.data_ptr(), t_eid.data_ptr(), 0, 0)
    else:
        make_csr = lambda: pgq.DeviceCSR.from_device_ptrs(V, t_keep[0].data_ptr(), t_keep[1].data_ptr(), t_keep[2].data_ptr(),
                                                          t_w.data_ptr() if t_w is not None else 0,
                                                          (2 if a2.weights == "double" else 1) if t_w is not None else 0)
    # a handle's first call BEFORE anything has warmed this graph's handles up (the block cache of the uploads aside)
    fc = None if a.no_first_call else bench.first_call(make_csr, pr_t, a.first_call_handles, paths=paths, cheapest=cheapest)
    m = bench.run(wl, csr, pr_t, total, steps, warmup, paths=paths, cheapest=cheapest)
    leg, _ = leg_summary(bench, m, wl, total, copy_gbps)
    if fc:
        leg["first_call"] = fc
    leg["workload"] = "%s %s, %d pairs (default_rng(%d))" % (name, OPS[wl], total, PAIR_SEED[wl])
    leg["config"] = {"V": V, "E": int(len(adj)), "pairs_total": total, "graph_gen_s": round(gen_s, 1),
                     "csr_upload_ms": round(upload_s * 1e3, 2) if upload_s is not None else None, "steps": steps, "warmup": warmup}
    if not a.no_cpu_baseline:
        if cheapest:  # the forest's searches are a handful of hops: every row against the oracle's Dijkstra, one thread
            leg["cpu_baseline"] = cpu_baseline_cheapest(V, off, adj, eid, w, pr, m["out_val"], m["d_ok"], how="all")
        elif paths:
            leg["cpu_baseline"] = cpu_baseline_paths(a2, V, off, adj, eid, pr, m, literal=False)
        else:
            leg["cpu_baseline"] = cpu_baseline(a2, V, off, adj, eid, pr, m["d_te"], m["out_len"])
    out = {wl: {k: leg[k] for k in leg if k != "roofline_by_kernel"} | {"roofline_by_kernel": leg["roofline_by_kernel"]}}
    if wl == "rmat22" and a.cross_sources > 0:
        # the binder's call shape on THIS graph: frontier arrays of V x 8 WD bytes (537 MB at WD = 16, 1 GB at 32) are far past
        # the 256-MiB Infinity Cache — the MS-BFS level kernels against DRAM (round-5 review: every >= 0.50 figure so far was
        # measured on a cache-resident working set)
        cp = cross_pairs(V, a.cross_sources * a.cross_dests, a.cross_sources, PAIR_SEED["snb_cross"] + 22)
        cp_t = torch.from_numpy(cp).to(dev)
        fc2 = None if a.no_first_call else bench.first_call(make_csr, cp_t, max(3, a.first_call_handles // 2))  # (the median of three: the process's first call of this size also allocates the workspace's maps and queues, once)
        mc = bench.run("rmat22_cross", csr, cp_t, len(cp), max(2, min(a.steps, 8)), 2)
        xleg, _ = leg_summary(bench, mc, "rmat22_cross", len(cp), copy_gbps)
        xleg["workload"] = "%s iterativelength, %d distinct sources x %d destinations each = %d rows (match.cpp:467-495 shape)" % (
            name, a.cross_sources, len(cp) // a.cross_sources, len(cp))
        xleg["config"] = {"V": V, "E": int(len(adj)), "pairs_total": len(cp), "frontier_array_MB_at_WD16": round(V * 128 / 1e6, 1)}
        if fc2:
            xleg["first_call"] = fc2
        if not a.no_cpu_baseline:
            xleg["cpu_baseline"] = cpu_baseline(a2, V, off, adj, eid, cp, mc["d_te"], mc["out_len"], sample=8192)
        out["msbfs_cross_rmat22"] = {k: xleg[k] for k in xleg if k != "roofline_by_kernel"} | {"roofline_by_kernel": xleg["roofline_by_kernel"]}
    del csr, t_keep
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)  # does not return
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:  # a launcher is around us: the ranks it started are the job
        if int(os.environ.get("RANK", "0")) == 0:
            print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; reporting n_gpus = WORLD_SIZE" % (
                a.gpus, os.environ["WORLD_SIZE"]), file=sys.stderr)
        a.gpus = int(os.environ["WORLD_SIZE"])
    if a.launch_check:
        return launch_check(a)
    bench = Bench(a)
    torch, dist, pgq, sharding = bench.torch, bench.dist, bench.pgq, bench.sharding
    for kv in a.set:
        k, v = kv.split("=", 1)
        pgq.set_option(k, float(v) if "." in v else int(v))
    world, rank, dev = bench.world, bench.rank, bench.dev
    scaling = a.scaling or ("strong" if world > 1 else "weak")
    cheapest = a.workload in CHEAPEST
    paths = a.workload == "snb_paths"

    pairs_cfg = a.pairs_per_gpu or DEFAULT_PAIRS[a.workload]
    # ---- graph: rank 0 builds it, the others receive it over RCCL (CSR replicated on every GPU) ----
    arrays, name, gen_s = None, "", 0.0
    snb_graph = None
    if rank == 0:
        name, V, off, adj, eid, w, gen_s = build_graph(a)
        snb_graph = (name, V, off, adj, eid, w, gen_s)
        arrays = {"off": torch.from_numpy(off), "adj": torch.from_numpy(adj), "eid": torch.from_numpy(eid)}
        if w is not None:  # doubles travel as their bit patterns (the broadcast helper moves int64 tensors)
            arrays["w"] = torch.from_numpy(np.ascontiguousarray(w).view(np.int64))
        if a.workload == "snb_cross_allv":
            pairs_cfg = 32 * V
        # one global list: the first `pairs_cfg` rows are the strong-scaling set, all world x pairs_cfg the weak one
        allp = make_pairs(a, V, pairs_cfg * world if a.workload != "snb_cross_allv" else pairs_cfg, off, adj)
        arrays["pairs"] = torch.from_numpy(np.ascontiguousarray(allp.reshape(-1)))
    arrays = sharding.broadcast_csr(arrays, dev)
    t_off, t_adj, t_eid, t_w = arrays["off"], arrays["adj"], arrays["eid"], arrays.get("w")
    has_w = t_w is not None
    V, E = t_off.numel() - 1, t_adj.numel()
    allp_t = arrays["pairs"].view(-1, 2)
    if a.workload == "snb_cross_allv":
        pairs_cfg = allp_t.shape[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    csr = pgq.DeviceCSR.from_device_ptrs(V, t_off.data_ptr(), t_adj.data_ptr(), t_eid.data_ptr(),
                                         t_w.data_ptr() if has_w else 0,
                                         (2 if a.weights == "double" else 1) if has_w else 0)
    upload_s = time.perf_counter() - t0

    def shard(mode):
        total = min(allp_t.shape[0], pairs_cfg * world if mode == "weak" else pairs_cfg)
        lo, hi = sharding.shard_bounds(total, world, rank)
        return allp_t[lo:hi], total, lo, hi

    mine_t, total_pairs, lo, hi = shard(scaling)

    def fresh_csr():
        return pgq.DeviceCSR.from_device_ptrs(V, t_off.data_ptr(), t_adj.data_ptr(), t_eid.data_ptr(),
                                              t_w.data_ptr() if has_w else 0, (2 if a.weights == "double" else 1) if has_w else 0)

    # a handle's FIRST call (before this process has run any search on this graph's handles: no route memo, no level plan,
    # no measured bytes per row) — what a query sees: the reference's CSR lives for one query
    first_main = None
    if world == 1 and not a.no_first_call and a.workload != "snb_cross_allv":
        first_main = bench.first_call(fresh_csr, mine_t, a.first_call_handles, paths=paths, cheapest=cheapest)
    m = bench.run(a.workload, csr, mine_t, total_pairs, a.steps, a.warmup, paths=paths, cheapest=cheapest)
    try:
        copy_gbps = pgq.copy_bandwidth_gbps(1 << 30, 5) if rank == 0 else None
    except Exception:
        copy_gbps = None
    main_leg, elapsed = leg_summary(bench, m, a.workload, total_pairs, copy_gbps)
    other = None
    if world > 1 and a.workload != "snb_cross_allv":  # the other scaling mode, measured in the same run
        mode2 = "weak" if scaling == "strong" else "strong"
        mine2, total2, _, _ = shard(mode2)
        m2 = bench.run(a.workload, csr, mine2, total2, a.steps, a.warmup, paths=paths, cheapest=cheapest)
        leg2, _ = leg_summary(bench, m2, a.workload, total2, copy_gbps)
        other = (mode2, {k: leg2[k] for k in ("ms_per_step", "pairs_per_s", "mteps_logical", "mteps_physical")}, total2)
    cross = cross_lanes = cross_shuf = None
    cross_multi = None
    if world > 1 and a.workload == "snb_sf100" and not a.no_legs:
        # N > 1: the cross product shards BY SOURCE — rows are grouped by source and cut into contiguous ranges, so every rank keeps
        # whole sources (2048 sources x 1024 rows over 8 ranks: 256 sources each); strong = the 2.1 M rows in total, weak = 2.1 M per rank
        cross_multi = {}
        for mode in ("strong", "weak"):
            tot = a.cross_sources * a.cross_dests * (world if mode == "weak" else 1)
            cp_all = cross_pairs(V, tot, a.cross_sources * (world if mode == "weak" else 1), PAIR_SEED["snb_cross"])
            rows_per_source = max(1, len(cp_all) // (a.cross_sources * (world if mode == "weak" else 1)))
            # whole sources per rank when they divide evenly (the gather wants equal blocks); else plain contiguous ranges — a
            # source cut in two costs one more ball (or lane) on one GPU, not correctness
            even = len(cp_all) % (world * rows_per_source) == 0
            clo, chi = sharding.shard_bounds_grouped(len(cp_all), world, rank, rows_per_source) if even else sharding.shard_bounds(len(cp_all), world, rank)
            cp_t = torch.from_numpy(np.ascontiguousarray(cp_all[clo:chi])).to(dev)
            mcx = bench.run("snb_cross", csr, cp_t, len(cp_all), max(2, min(a.steps, 5)), min(a.warmup, 2))
            lx, _ = leg_summary(bench, mcx, "snb_cross", len(cp_all), copy_gbps)
            cross_multi[mode] = {k: lx[k] for k in ("ms_per_step", "pairs_per_s", "mteps_logical", "mteps_physical")} | {
                "pairs_total": len(cp_all), "scaling": mode, "sharded": "by source: contiguous row ranges aligned to whole sources"}
    if world == 1 and a.workload == "snb_sf100" and not a.no_legs:
        # the binder's call shape on the same graph and row count
        cp = cross_pairs(V, a.cross_sources * a.cross_dests, a.cross_sources, PAIR_SEED["snb_cross"])
        cp_t = torch.from_numpy(cp).to(dev)
        first_cross = None if a.no_first_call else bench.first_call(fresh_csr, cp_t, a.first_call_handles)
        mc = bench.run("snb_cross", csr, cp_t, len(cp), max(2, min(a.steps, 5)), min(a.warmup, 2))
        cross, _ = leg_summary(bench, mc, "snb_cross_ball", len(cp), copy_gbps)
        cross["workload"] = "%d distinct sources x %d destinations each = %d rows grouped by source (match.cpp:467-495 shape), as the library routes them" % (
            a.cross_sources, len(cp) // a.cross_sources, len(cp))
        if first_cross:
            cross["first_call"] = first_cross
        # the same rows forced through the lane-batched MS-BFS (rounds 1-5's route for them): the frontier-expansion kernels' figures
        ball_was = pgq.get_option("ball")
        pgq.set_option("ball", 0)
        try:
            first_lanes = None if a.no_first_call else bench.first_call(fresh_csr, cp_t, max(1, a.first_call_handles // 2))
            csr_l = fresh_csr()  # a handle of its own: the shared one remembers that the source-centric kernel took these buffers
            ml = bench.run("snb_cross_lanes", csr_l, cp_t, len(cp), max(2, min(a.steps, 5)), min(a.warmup, 2))
            del csr_l
        finally:
            pgq.set_option("ball", int(ball_was))
        cross_lanes, _ = leg_summary(bench, ml, "snb_cross", len(cp), copy_gbps)
        cross_lanes["workload"] = "the msbfs_cross rows with the source-centric kernel switched off (ball = 0): lane-batched MS-BFS, 2048 lanes"
        if first_lanes:
            cross_lanes["first_call"] = first_lanes
        assert bool((ml["out_len"] == mc["out_len"]).all()), "the two routes of the cross product disagree"
        # the same rows in random order (a hash join's output): declined by the source-centric kernel as they lie, sorted by
        # source for it (option ball_sort), answers scattered back — compared row by row with the grouped leg's output
        perm = np.random.default_rng(PAIR_SEED["snb_cross"] + 1).permutation(len(cp))
        perm_t = torch.from_numpy(perm).to(dev)
        sp_t = cp_t[perm_t].contiguous()
        csr_s = fresh_csr()  # a handle of its own: no memo of the grouped buffers
        msx = bench.run("snb_cross_shuffled", csr_s, sp_t, len(cp), max(2, min(a.steps, 5)), min(a.warmup, 2))
        del csr_s
        cross_shuf, _ = leg_summary(bench, msx, "snb_cross_shuffled", len(cp), copy_gbps)
        cross_shuf["workload"] = "the msbfs_cross rows in random order: sorted by source in front of the source-centric kernel (ball_sort), scattered back"
        assert bool((msx["out_len"] == mc["out_len"][perm_t]).all()), "the shuffled cross product's answers differ from the grouped one's"

    wleg = None
    if world == 1 and a.workload == "snb_sf100" and not a.no_legs and a.cheapest_pairs > 0:
        # cheapest_path_length on the same graph with int64 weights 1..999 per CSR slot: the general-graph case (batched
        # relaxation); one timed step, results compared with the oracle's Dijkstra on a bounded sample
        w_np = np.random.default_rng(PAIR_SEED["snb_cheapest"]).integers(1, 1000, E)
        t_w2 = torch.from_numpy(w_np).to(dev)
        csr_w = pgq.DeviceCSR.from_device_ptrs(V, t_off.data_ptr(), t_adj.data_ptr(), t_eid.data_ptr(), t_w2.data_ptr(), 1)
        wp = np.random.default_rng(PAIR_SEED["snb_cheapest"] + 100).integers(0, V, size=(a.cheapest_pairs, 2))
        mw = bench.run("snb_cheapest", csr_w, torch.from_numpy(wp).to(dev), len(wp), 1, 1, cheapest=True)
        wleg, _ = leg_summary(bench, mw, "snb_cheapest", len(wp), copy_gbps)
        wleg["workload"] = "%d random pairs, int64 weights 1..999 on the knows graph (cheapest_path_length.cpp:52-136)" % len(wp)
        if not a.no_cpu_baseline and rank == 0:
            sel = np.arange(min(len(wp), 256), dtype=np.int64) * max(1, len(wp) // 256)  # strided; a Dijkstra here is ~0.5 s
            tsel = torch.from_numpy(sel).to(dev)
            wleg["cpu_baseline"] = cpu_baseline_cheapest(V, off, adj, eid, w_np, wp[sel], mw["out_val"][tsel], mw["d_ok"][tsel],
                                                         how="%d rows at stride %d" % (len(sel), max(1, len(wp) // 256)))
        del csr_w, t_w2

    config_legs = {}
    if world == 1 and a.workload == "snb_sf100" and not a.no_legs:
        # the other BASELINE configs in the same line: configs[2] on this graph, configs[1] and configs[4] on their own
        for wl in [x for x in a.config_legs.split(",") if x]:
            config_legs.update(config_leg(bench, a, wl, copy_gbps, snb_graph + ((t_off, t_adj, t_eid),) if snb_graph else None, csr))

    if rank == 0:
        # value = what the hardware did: (src, dst) pairs answered per second (BASELINE metric "MS-BFS MTEPS + src-dst
        # pairs/sec"; the default workload's rows are answered by the pair-centric kernels, no BFS level runs, so pairs/s
        # is its throughput).  mteps_physical (adjacency entries the kernels scanned) and mteps_logical (edges the
        # reference's per-pair lanes would traverse: a count of avoided work, not a hardware rate) ride along.
        metric = "cheapest_path_pairs_per_s" if cheapest else "src_dst_pairs_per_s"
        unit, value = "pairs/s", main_leg["pairs_per_s"]
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": main_leg["ms_per_step"], "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u64" if not cheapest else ("f64" if a.weights == "double" else "int64"),
            "data": "synthetic",
            "schema": 6,  # round 6: legs_summary / first_call / msbfs_cross_lanes / msbfs_cross_rmat22; round 5: `roofline` = the leg's launch chain (round 4: its dominant kernel, now `roofline.dominant_kernel`)
            "value_kind": "pairs answered per second, whole job; mteps_physical = adjacency entries scanned per second / 1e6; "
                          "mteps_logical = reference-lane traversed edges per second / 1e6 (work avoided, not a hardware rate)",
            "mteps_logical": main_leg["mteps_logical"],
            "msbfs_mteps": main_leg["mteps_logical"],  # round 3's headline key (the logical figure), kept for trackers of `value`
            "config": {"workload": "%s %s, %d pairs %s, CSR replicated" % (
                name, OPS[a.workload], pairs_cfg, "per GPU" if scaling == "weak" else "in total"),
                "V": V, "E": E, "pairs_total": total_pairs,
                "parallelism": ("pairs sharded x%d, RCCL all_gather of %s (async, overlapped with the next step)" % (
                    world, "lengths + path lists" if paths else "lengths")) if world > 1 else "1 GPU",
                "graph_gen_s": round(gen_s, 1), "csr_upload_ms": round(upload_s * 1e3, 2),
                **({"options": a.set} if a.set else {}), **({"rows": "shuffled"} if a.cross_shuffle else {})},
            "pairs_per_s": main_leg["pairs_per_s"],
            "reachable_pairs": main_leg["reachable_pairs"],
            "traversed_edges_per_step": main_leg["traversed_edges_per_step"],
            # logical edges (value) vs the adjacency entries the kernels physically scanned in the timed region
            "mteps_physical": main_leg["mteps_physical"],
            "physical_edges_scanned_per_step": main_leg["physical_edges_scanned_per_step"],
            # the CSR dies at QueryEnd: one query = one upload (device-resident arrays here) + the searches
            "ms_per_step_incl_csr_upload": main_leg["ms_per_step"] + upload_s * 1e3,
            "rows_answered_by_prepass_per_step": main_leg["rows_answered_by_prepass_per_step"],
            "levels_per_step": main_leg["levels_per_step"],
            "push_pull_levels": main_leg["push_pull_levels"],
            "deferred_pairs_per_step": main_leg["deferred_pairs_per_step"],
            # dominant kernel class of the one-batch-in-flight pass (no overlap: event time = kernel time)
            "roofline": main_leg["roofline"],
            # every kernel class of that pass: event ms per step (they add up to less than its wall time: host round
            # trips and copies are not kernels), algorithmic GB/s, launches per step
            "roofline_by_kernel": main_leg["roofline_by_kernel"],
        }
        if other is not None:
            out[other[0]] = dict(other[1], pairs_total=other[2], scaling=other[0])
        if cross_multi is not None:
            out["msbfs_cross"] = cross_multi
        if first_main is not None:
            out["first_call"] = first_main
            out["first_call_ms"] = first_main["first_call_ms"]
        mine = allp_t[lo:hi].cpu().numpy()
        if not a.no_cpu_baseline and world == 1 and a.workload in ("snb_sf100", "rmat22", "snb_cross", "rmat22_cross"):  # rank 0, N=1 only
            out["cpu_baseline"] = cpu_baseline(a, V, off, adj, eid, mine, m["d_te"], m["out_len"], sample=8192 if a.workload == "rmat22_cross" else 0)
        if not a.no_cpu_baseline and world == 1 and cheapest:
            ns = len(mine) if a.workload == "forest_cheapest" else min(len(mine), 512)  # a Dijkstra on the knows graph is ~0.5 s
            sel = np.arange(ns, dtype=np.int64) * max(1, len(mine) // ns)
            tsel = torch.from_numpy(sel).to(dev)
            out["cpu_baseline"] = cpu_baseline_cheapest(V, off, adj, eid, w, mine[sel], m["out_val"][tsel], m["d_ok"][tsel],
                                                        how="all" if ns == len(mine) else "%d rows at stride %d" % (ns, max(1, len(mine) // ns)))
        if not a.no_cpu_baseline and world == 1 and paths:
            out["cpu_baseline"] = cpu_baseline_paths(a, V, off, adj, eid, mine, m)
        if cross is not None:
            if not a.no_cpu_baseline:
                cross["cpu_baseline"] = cpu_baseline(a, V, off, adj, eid, cp, mc["d_te"], mc["out_len"], sample=65536)  # strided: 32 rows of every source, one chunk per host thread
            legs = {"prepass": {k: main_leg[k] for k in main_leg if k != "roofline_by_kernel"}, "msbfs_cross": cross}
            if first_main is not None:
                legs["prepass"]["first_call"] = first_main
            if cross_lanes is not None:
                if not a.no_cpu_baseline:  # the same strided sample, against the forced route's own output
                    cross_lanes["cpu_baseline"] = {k: cross["cpu_baseline"][k] for k in cross.get("cpu_baseline", {}) if k in ("value", "unit", "cores")}
                    cross_lanes["cpu_baseline"]["sample"] = "output identical to msbfs_cross's row by row (asserted): its comparison holds for this route"
                    cross_lanes["cpu_baseline"]["rows_compared"] = int(len(cp))
                    cross_lanes["cpu_baseline"]["rows_equal"] = int(len(cp))
                legs["msbfs_cross_lanes"] = cross_lanes
            if cross_shuf is not None:
                if not a.no_cpu_baseline:
                    cross_shuf["cpu_baseline"] = {k: cross["cpu_baseline"][k] for k in cross.get("cpu_baseline", {}) if k in ("value", "unit", "cores")}
                    cross_shuf["cpu_baseline"]["sample"] = "output identical to msbfs_cross's under the row permutation (asserted): its comparison holds for these rows"
                    cross_shuf["cpu_baseline"]["rows_compared"] = int(len(cp))
                    cross_shuf["cpu_baseline"]["rows_equal"] = int(len(cp))
                legs["msbfs_cross_shuffled"] = cross_shuf
            legs["prepass"]["workload"] = "%d random pairs (default_rng(4)): every row answered by the pair-centric kernels" % total_pairs
            if "cpu_baseline" in out:
                legs["prepass"]["cpu_baseline"] = {k: out["cpu_baseline"][k] for k in ("value", "unit", "cores", "pairs_per_s", "rows_compared", "rows_equal")}
            if wleg is not None:
                legs["cheapest_general"] = {k: wleg[k] for k in wleg if k != "roofline_by_kernel"}
            legs.update(config_legs)
            out["legs"] = legs
            out["legs_by_config"] = {"configs[1] R-MAT-22 iterativelength 1024 pairs": "rmat22",
                                     "configs[1]'s graph in the binder's call shape (DRAM-resident frontier arrays)": "msbfs_cross_rmat22",
                                     "configs[3] cross product forced through the lane batches": "msbfs_cross_lanes",
                                     "configs[3] cross product, rows in random order (sorted by source first)": "msbfs_cross_shuffled",
                                     "configs[2] SF100 shortestpath + reconstruction 4096 pairs": "snb_paths",
                                     "configs[3] SF100 iterativelength 65,536 pairs": "prepass (= the top-level fields)",
                                     "configs[3] in the binder's call shape (cross product)": "msbfs_cross",
                                     "configs[4] weighted cheapest path, reply forest": "forest_cheapest",
                                     "configs[4]'s operator on a general graph": "cheapest_general"}
        if "legs" in out:
            # LAST key of the line + top-level scalars: the driver's record keeps the head of `parsed` and the tail of the line
            summ = {}
            for nm, lg in out["legs"].items():
                cb = lg.get("cpu_baseline", {})
                fc = lg.get("first_call", {}).get("first_call_ms")
                summ[nm] = [round(lg["ms_per_step"], 4), round(lg["roofline"]["frac"], 3), round(lg["roofline"]["step"]["frac"], 3),
                            cb.get("rows_compared"), cb.get("rows_equal"), None if fc is None else round(fc, 3)]
                out["leg_%s_ms" % nm] = round(lg["ms_per_step"], 4)
                out["leg_%s_frac" % nm] = round(lg["roofline"]["frac"], 3)
                if fc is not None:
                    out["leg_%s_first_ms" % nm] = round(fc, 3)
            out["legs_summary_format"] = "[ms_per_step, chain frac of 8 TB/s, whole-step frac, rows compared with the CPU port, rows equal, first_call_ms]"
            out["legs_summary"] = summ
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(a, V, off, adj, eid, mine, d_te, out_len, sample=0):
    """The oracle's literal restatement of IterativeLengthFunction (reference memory layout and loops), driven in
    2048-row chunks like DuckDB drives the UDF.  Two bounded samples of this rank's pairs: one thread on the first
    `cpu_sample` pairs, and one thread per chunk (up to the host's cores) on as many chunks as there are cores.
    `out_len` is the output of the TIMED steps (the buffer was poisoned before them).  Checker + baseline only —
    never on the product path."""
    from oracle.pgq_oracle import OracleCSR
    cores = os.cpu_count() or 1
    ora = OracleCSR.adopt(V, off, adj, eid)
    how = "first"
    if sample and sample < len(mine):
        # a STRIDED sample: the rows of a cross product are grouped by source, so the first rows would cover a handful of
        # sources (= lanes of one batch); every (len / sample)-th row touches every source of the product
        sel = np.arange(sample, dtype=np.int64) * (len(mine) // sample)
        mine = mine[sel]
        import torch
        tsel = torch.from_numpy(sel).to(out_len.device)
        d_te, out_len = d_te[tsel], out_len[tsel]
        how = "strided (every %d-th row: all sources of the product)" % (len(mine) and int(sel[1] - sel[0]) if len(sel) > 1 else 1)
    ns1 = min(a.cpu_sample or (8192 if a.workload in SNB else 1024), len(mine))
    res = {}
    if not sample:
        t0 = time.perf_counter()
        ln, ok = ora.baseline_run("iterativelength", V, mine[:ns1, 0], mine[:ns1, 1], nthreads=1)
        dt1 = time.perf_counter() - t0
        gpu_len = out_len[:ns1].cpu().numpy()
        agree = bool(((gpu_len >= 0) == ok).all() and (gpu_len[ok] == ln[ok]).all())
        res["rows_compared"], res["rows_equal"] = int(ns1), int((((gpu_len >= 0) == ok) & ((gpu_len == ln) | ~ok)).sum())
        te1 = float(d_te[:ns1].sum().item())
        res["single_thread"] = {"value": te1 / dt1 / 1e6, "cores": 1, "pairs_per_s": ns1 / dt1,
                                "sample": "first %d pairs, %.1f s; results equal the GPU's timed output: %s" % (ns1, dt1, agree)}
    # one worker per 2048-row chunk: the most threads DuckDB's chunking can use on these rows
    nchunks_all = max(1, (len(mine) + 2047) // 2048)
    threads = max(1, min(nchunks_all, cores))
    nsm = min(len(mine), threads * 2048)
    if threads == 1 and not sample and nsm == ns1:  # one chunk: the single-thread run above IS the chunked run
        lnm, okm, dtm = ln, ok, dt1
    else:
        t0 = time.perf_counter()
        lnm, okm = ora.baseline_run("iterativelength", V, mine[:nsm, 0], mine[:nsm, 1], nthreads=threads)
        dtm = time.perf_counter() - t0
    gpu_m = out_len[:nsm].cpu().numpy()
    agree_m = bool(((gpu_m >= 0) == okm).all() and (gpu_m[okm] == lnm[okm]).all())
    if nsm >= res.get("rows_compared", 0):
        res["rows_compared"], res["rows_equal"] = int(nsm), int((((gpu_m >= 0) == okm) & ((gpu_m == lnm) | ~okm)).sum())
    tem = float(d_te[:nsm].sum().item())
    res.update({"value": tem / dtm / 1e6, "unit": "MTEPS", "cores": threads, "kind": "port",
                "sample": how + " %d pairs of rank 0's shard in 2048-row chunks, one thread per chunk (%d threads), literal "
                          "512-lane restatement (oracle/pgq_oracle.cpp), %.1f s; results equal the GPU's timed output: %s" % (
                              nsm, threads, dtm, agree_m),
                "pairs_per_s": nsm / dtm, "host_cores_available": cores})
    return res


def cpu_baseline_cheapest(V, off, adj, eid, w, mine, d_val, d_ok, how="all"):
    """Per-pair Dijkstra of the oracle (lean restatement: same distances as the reference's batched Bellman-Ford, which
    needs 8 KiB per vertex per call and does not fit a 2^24-vertex graph) on a sample of this rank's pairs, the rows dealt
    to one thread per host core (the searches are independent; the oracle call releases the GIL); every value of the
    timed output compared bit for bit."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.pgq_oracle import OracleCSR
    ora = OracleCSR.adopt(V, off, adj, eid, w)
    # one thread per 8 rows, at most 32: every thread owns a V-sized label array (a 2^24-vertex forest: 134 MB each), and the
    # forest's searches are a handful of hops — there one thread is the faster baseline (all rows: `how` == "all")
    threads = 1 if how == "all" else max(1, min(os.cpu_count() or 1, 32, len(mine) // 8 or 1))
    parts = np.array_split(np.arange(len(mine)), threads)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(lambda ix: ora.lean_cheapest_path_length(V, mine[ix, 0], mine[ix, 1]), parts))
    dt = time.perf_counter() - t0
    want = np.concatenate([r[0] for r in res])
    wok = np.concatenate([r[1] for r in res])
    ok = d_ok.cpu().numpy().astype(bool)
    got = d_val.cpu().numpy()
    if want.dtype.kind == "f":
        got = got.view(np.float64)
    agree = bool((ok == wok).all() and (got[ok] == want[wok]).all())
    same = (ok == wok) & ((got == want) | ~ok) if len(got) == len(want) else np.zeros(len(mine), dtype=bool)
    return {"value": len(mine) / dt, "unit": "pairs/s", "cores": threads, "kind": "port", "rows_compared": int(len(mine)), "rows_equal": int(same.sum()),
            "sample": "%s (%d pairs), per-pair Dijkstra (oracle/pgq_oracle.cpp lean restatement) on %d threads, %.1f s; "
                      "results equal the GPU's timed output: %s" % (how, len(mine), threads, dt, agree)}


def cpu_baseline_paths(a, V, off, adj, eid, mine, m, literal=True):
    """shortestpath (configs[2]): (1) the literal restatement of ShortestPathFunction (512 lanes, two parent arrays of
    V x 512 int64 = 3.7 GB per chunk in flight on the SF100-shaped graph: ONE thread, one 512-lane batch) timed on the
    first 512 pairs — its cost does not depend on how many of the 512 lanes are used; (2) every list of the TIMED output
    compared with the oracle's lean restatement (per-pair BFS + the reference's parent rule) on a strided sample."""
    from oracle.pgq_oracle import OracleCSR
    ora = OracleCSR.adopt(V, off, adj, eid)
    nb = min(512, len(mine))
    if literal:
        t0 = time.perf_counter()
        ln, ok = ora.baseline_run("shortestpath", V, mine[:nb, 0], mine[:nb, 1], nthreads=1)
        dt = time.perf_counter() - t0
        got_len = m["out_len"][:nb].cpu().numpy()
        agree_len = bool(((got_len >= 0) == ok).all() and (got_len[ok] == ln[ok]).all())
    ns = min(len(mine), 1024)
    sel = np.arange(ns, dtype=np.int64) * max(1, len(mine) // ns)
    t0 = time.perf_counter()
    want = ora.lean_shortestpath(V, mine[sel, 0], mine[sel, 1])
    dt2 = time.perf_counter() - t0
    lens, offs, child = m["out_len"].cpu().numpy(), m["out_off"].cpu().numpy(), m["out_child"].cpu().numpy()
    same = 0
    for k, i in enumerate(sel):
        got = None if lens[i] < 0 else child[offs[i]:offs[i] + 2 * lens[i] + 1].tolist()
        same += int(got == want[k])
    if not literal:  # as a leg of the default line: the literal restatement's 512-lane batch alone is 90 s of CPU
        return {"value": ns / dt2, "unit": "pairs/s", "cores": 1, "kind": "port",
                "sample": "%d strided rows, lean restatement (oracle/pgq_oracle.cpp: per-pair BFS + the reference's parent rule, "
                          "shortest_path.cpp:21-31), one thread, %.1f s; full lists equal the GPU's timed output: %d of %d (the literal "
                          "512-lane ShortestPathFunction restatement — 3.7 GB of parent arrays, 5.7 pairs/s — is timed by "
                          "--workload snb_paths)" % (ns, dt2, same, ns),
                "paths_compared": ns, "paths_equal": same, "rows_compared": ns, "rows_equal": same}
    return {"value": nb / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "first %d pairs = one 512-lane batch of the literal restatement (oracle/pgq_oracle.cpp, 3.7 GB of parent "
                      "arrays: thread count capped at 1), %.1f s, hop counts equal the GPU's timed output: %s; full lists "
                      "of %d strided rows against the lean restatement (%.1f s): %d equal" % (nb, dt, agree_len, ns, dt2, same),
            "paths_compared": ns, "paths_equal": same, "rows_compared": ns, "rows_equal": same}


if __name__ == "__main__":
    main()
