#!/usr/bin/env python3
"""bench.py — DuckPGQ's path-finding hot path on MI355X (contract: one JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over this rank's batch of (src,dst) pairs, inputs resident in HBM:
pgq_iterativelength_bulk_device (pair-centric pre-pass for the pairs at distance <= 4, lane-batched MS-BFS for the
rest), followed for N > 1 by the RCCL all_gather of the per-pair results (the only inter-GPU traffic; the CSR is
replicated).

Default workload = BASELINE.json configs[3] (the config the metric "MS-BFS MTEPS + src-dst pairs/sec, SNB SF100,
1/2/4/8 GPU" is quoted on; it fits one GPU): synthetic LDBC-SNB-SF100-shaped Person-knows-Person graph (V=448,626,
39.88 M symmetric CSR entries), iterativelength on 65,536 random pairs per GPU (`default_rng(4)`).  --scaling weak
(default): every rank gets its own 65,536 pairs of one global list; --scaling strong: the 65,536 pairs are cut across
the ranks.  Other BASELINE configs: --workload rmat22 (configs[1]), snb_paths (configs[2]), forest_cheapest (configs[4]).

value    = MTEPS: traversed edges / second / 1e6, summed over ranks.  Traversed edges of a pair = out-degrees of all
           vertices its own level-synchronous BFS expands up to the level that reaches dst (all levels if unreachable)
           — a pure function of (graph, src, dst), counted once on the GPU outside the timed region
           (pgq_traversed_edges_bulk_device) and pinned against the CPU oracle in tests/.
roofline = the dominant kernel class of an untimed pass with one batch in flight and per-launch HIP events on the
           library's own stream (no overlap: a launch's event duration is its own duration); achieved = algorithmic
           bytes / that time (DESIGN.md has the formulas).  `step` = all kernel classes' algorithmic bytes over the
           wall time of the timed region.
cpu_baseline = the literal restatement of the reference UDF (oracle/, 512-lane bitsets, 2048-row chunks) timed on this
           box's host cores on a bounded sample of the same pairs, same MTEPS definition, one thread and one thread
           per chunk.
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

OPS = {"snb_sf100": "iterativelength", "rmat22": "iterativelength", "snb_paths": "shortestpath+reconstruction",
       "forest_cheapest": "cheapest_path_length", "snb_cheapest": "cheapest_path_length"}
DEFAULT_PAIRS = {"snb_sf100": 65536, "rmat22": 1024, "snb_paths": 4096, "forest_cheapest": 4096, "snb_cheapest": 4096}
PAIR_SEED = {"snb_sf100": 4, "rmat22": 2, "snb_paths": 3, "forest_cheapest": 5, "snb_cheapest": 6}
CHEAPEST = ("forest_cheapest", "snb_cheapest")  # weighted workloads: value = pairs/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="snb_sf100", choices=sorted(OPS))
    ap.add_argument("--pairs-per-gpu", type=int, default=0, help="0 = the BASELINE config's pair count")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: pairs-per-gpu rows on every rank; strong: pairs-per-gpu rows in total, cut across ranks")
    ap.add_argument("--scale", type=int, default=0, help="override graph scale (rmat scale / forest log2 V); tests")
    ap.add_argument("--snb-vertices", type=int, default=448626)
    ap.add_argument("--snb-friendships", type=int, default=19_940_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs timed on one CPU thread (0 = 8192 snb / 1024 rmat)")
    ap.add_argument("--weights", default="int64", choices=["int64", "double"], help="forest_cheapest / snb_cheapest: weight type")
    ap.add_argument("--backend", default="nccl")
    return ap.parse_args()


def build_graph(a):
    from duckpgq_extension_amd import graphgen
    t0 = time.time()
    w = None
    if a.workload in ("snb_sf100", "snb_paths", "snb_cheapest"):
        V, s, d = graphgen.snb_knows_like(a.snb_vertices, a.snb_friendships, seed=100)
        name = "snb_sf100_knows(V=%d)" % V
        if a.workload == "snb_cheapest":  # the general-graph case of cheapest_path_length: weights 1..999 on the knows graph
            w = np.random.default_rng(6).integers(1, 1000, len(s))
            if a.weights == "double":
                w = w.astype(np.float64) / 7.0
            name += ",%s w" % a.weights
    elif a.workload == "rmat22":
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


def make_pairs(a, V, total, off, adj):
    """One global, seeded pair list.  The reply forest gets destinations that are ancestors of their sources (uniform
    pairs are almost never connected there: the search would only measure the dead-end shortcut)."""
    rng = np.random.default_rng(PAIR_SEED[a.workload])
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


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    import duckpgq_extension_amd as pgq
    from duckpgq_extension_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=a.backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % ndev)
    dev = torch.device("cuda", local % ndev)
    pgq.load_hip().pgq_init(local % ndev)

    pairs_per_gpu = a.pairs_per_gpu or DEFAULT_PAIRS[a.workload]
    total_pairs = pairs_per_gpu * world if a.scaling == "weak" else pairs_per_gpu
    # ---- graph: rank 0 builds it (cached on disk), the others receive it over RCCL (CSR replicated on every GPU) ----
    arrays, name, gen_s = None, "", 0.0
    if rank == 0:
        name, V, off, adj, eid, w, gen_s = build_graph(a)
        arrays = {"off": torch.from_numpy(off), "adj": torch.from_numpy(adj), "eid": torch.from_numpy(eid)}
        if w is not None:  # doubles travel as their bit patterns (the broadcast helper moves int64 tensors)
            arrays["w"] = torch.from_numpy(np.ascontiguousarray(w).view(np.int64))
        allp = make_pairs(a, V, total_pairs, off, adj)
        arrays["pairs"] = torch.from_numpy(np.ascontiguousarray(allp.reshape(-1)))
    arrays = sharding.broadcast_csr(arrays, dev)
    t_off, t_adj, t_eid, t_w = arrays["off"], arrays["adj"], arrays["eid"], arrays.get("w")
    has_w = t_w is not None
    V, E = t_off.numel() - 1, t_adj.numel()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    csr = pgq.DeviceCSR.from_device_ptrs(V, t_off.data_ptr(), t_adj.data_ptr(), t_eid.data_ptr(),
                                         t_w.data_ptr() if has_w else 0,
                                         (2 if a.weights == "double" else 1) if has_w else 0)
    upload_s = time.perf_counter() - t0

    # ---- pairs: one global list, contiguous shard per rank ------------------------------------------------------
    lo, hi = sharding.shard_bounds(total_pairs, world, rank)
    mine_t = arrays["pairs"].view(-1, 2)[lo:hi]
    n = hi - lo
    d_src, d_dst = mine_t[:, 0].contiguous(), mine_t[:, 1].contiguous()
    d_len = torch.empty(n, dtype=torch.int64, device=dev)
    d_te = torch.zeros(n, dtype=torch.int64, device=dev)
    child_cap = n * 64
    d_off = d_child = d_val = d_ok = None
    if a.workload == "snb_paths":
        d_off = torch.zeros(n, dtype=torch.int64, device=dev)
        d_child = torch.empty(child_cap, dtype=torch.int64, device=dev)
    if a.workload in CHEAPEST:
        d_val = torch.zeros(n, dtype=torch.int64, device=dev)
        d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    used_box = [0]

    def step():
        if a.workload == "snb_paths":
            rc, used = csr.shortestpath_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr(),
                                                 d_off.data_ptr(), d_child.data_ptr(), child_cap)
            assert rc == 0, pgq.load_hip().pgq_last_error()
            used_box[0] = used
        elif a.workload in CHEAPEST:
            csr.cheapest_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_val.data_ptr(), d_ok.data_ptr())
        else:
            csr.iterativelength_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr())
        if world > 1:  # final RCCL gather of the per-pair results (xGMI): lengths, and the path lists for shortestpath
            per = (total_pairs + world - 1) // world
            if a.workload == "snb_paths":
                sharding.gather_paths(d_len, d_off, d_child, used_box[0], per)
            else:
                sharding.gather_rows(d_val if a.workload in CHEAPEST else d_len, total_pairs)

    # ---- work units (outside the timed region) -----------------------------------------------------------------
    if a.workload not in CHEAPEST:
        csr.traversed_edges_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr(), d_te.data_ptr())
    te_local = int(d_te.sum().item())
    ref_len = d_len.clone()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    pgq.set_option("profile", 0)
    pgq.reset_stats()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    stats = pgq.get_stats()
    if a.workload in CHEAPEST:
        reach = int(d_ok.sum().item())
    else:
        assert bool((d_len == ref_len).all()), "results differ from the traversed-edge accounting pass"
        reach = int((d_len >= 0).sum().item())
    # Untimed pass with one batch in flight and HIP events around every launch (recorded on the library's own stream):
    # nothing overlaps, so a launch's event duration is that kernel's duration.  `value` is the timed region above.
    n_streams = int(pgq.get_option("streams"))
    pgq.set_option("streams", 1)
    pgq.set_option("profile", 1)
    iso_steps = max(1, min(a.steps, 3))
    step()
    pgq.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iso_steps):
        step()
    torch.cuda.synchronize()
    iso_elapsed = time.perf_counter() - t0
    iso = pgq.get_stats()
    pgq.set_option("profile", 0)
    pgq.set_option("streams", n_streams)

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(te_local), float(stats["edges_scanned"]), float(reach)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(el[0])
    te_total = float(tot[0])

    if rank == 0:
        kms, kb, kl = iso["kernel_ms"], iso["algo_bytes"], iso["launches"]
        dom = max(kms, key=lambda k: kms[k])
        ach = kb[dom] / 1e9 / (kms[dom] / 1e3) if kms[dom] > 0 else 0.0
        try:
            copy_gbps = pgq.copy_bandwidth_gbps(1 << 30, 5)
        except Exception:
            copy_gbps = None
        pairs_per_s = total_pairs * a.steps / elapsed
        if a.workload in CHEAPEST:
            metric, unit = "cheapest_path_pairs_per_s", "pairs/s"
            value = pairs_per_s
        else:
            metric, unit = "msbfs_mteps", "MTEPS"
            value = te_total * a.steps / elapsed / 1e6
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_%s.json" % a.workload)
        if os.path.exists(pmc):  # written by tools/pmc_summary.py from separate rocprofv3 --pmc passes of this command
            try:
                traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
                traffic_src = "profiles/pmc_%s.json (separate rocprofv3 --pmc passes, committed; not collected in this run)" % a.workload
            except Exception:
                traffic = None
        step_bytes = sum(stats["algo_bytes"].values())
        traffic_gbps = None  # the PMC traffic (separate passes) over this run's launch duration: what the memory system moved
        try:
            if traffic and kms[dom] > 0:
                traffic_gbps = float(traffic) / (kms[dom] / max(kl[dom], 1) * 1e-3) / 1e9
        except Exception:
            traffic_gbps = None
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "u64" if a.workload not in CHEAPEST else ("f64" if a.weights == "double" else "int64"),
            "data": "synthetic",
            "config": {"workload": "%s %s, %d pairs %s, CSR replicated" % (
                name, OPS[a.workload], pairs_per_gpu, "per GPU" if a.scaling == "weak" else "in total"),
                "V": V, "E": E, "pairs_total": total_pairs,
                "parallelism": ("pairs sharded x%d, RCCL all_gather of %s" % (
                    world, "lengths + path lists" if a.workload == "snb_paths" else "lengths")) if world > 1 else "1 GPU",
                "graph_gen_s": round(gen_s, 1), "csr_upload_ms": round(upload_s * 1e3, 2)},
            "pairs_per_s": pairs_per_s,
            "reachable_pairs": int(tot[2]),
            "traversed_edges_per_step": te_total,
            # logical edges (value) vs the adjacency entries the kernels physically scanned in the timed region
            "mteps_physical": float(tot[1]) / elapsed / 1e6,
            "physical_edges_scanned_per_step": float(tot[1]) / a.steps,
            # the CSR dies at QueryEnd: one query = one upload (device-resident arrays here) + the searches
            "ms_per_step_incl_csr_upload": elapsed / a.steps * 1e3 + upload_s * 1e3,
            "rows_answered_by_prepass_per_step": stats["meet_pairs"] / max(a.steps, 1),
            "levels_per_step": stats["levels"] / max(a.steps, 1),
            "push_pull_levels": [stats["push_levels"] // max(a.steps, 1), stats["pull_levels"] // max(a.steps, 1)],
            "deferred_pairs_per_step": stats["deferred_pairs"] / max(a.steps, 1),
            # dominant kernel class of the one-batch-in-flight pass (no overlap: event time = kernel time)
            "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_GBps": traffic_gbps,
                         "traffic_frac": (traffic_gbps / HBM_PEAK_GBPS) if traffic_gbps else None,
                         "launches": int(kl[dom]), "avg_launch_ms": kms[dom] / max(kl[dom], 1),
                         "algorithmic_bytes_per_launch": kb[dom] / max(kl[dom], 1),
                         "measured_copy_GBps": copy_gbps, "timing": "HIP events, one batch in flight, untimed pass",
                         "ms_per_step_of_that_pass": iso_elapsed / iso_steps * 1e3,
                         # all kernel classes' algorithmic bytes over the wall time of the timed region
                         "step": {"algorithmic_bytes": step_bytes / a.steps,
                                  "GBps": step_bytes / 1e9 / elapsed, "frac": step_bytes / 1e9 / elapsed / HBM_PEAK_GBPS}},
            # every kernel class of that pass: event ms per step (they add up to less than its wall time: host round
            # trips and copies are not kernels), algorithmic GB/s, launches per step
            "roofline_by_kernel": {k: {"ms_per_step": round(kms[k] / iso_steps, 4),
                                       "GBps": round(kb[k] / 1e9 / (kms[k] / 1e3), 1) if kb[k] > 0 else None,
                                       "launches_per_step": kl[k] / iso_steps}
                                   for k in kms if kms[k] > 0},
        }
        if not a.no_cpu_baseline and world == 1 and a.workload in ("snb_sf100", "rmat22"):  # rank 0, N=1 only
            mine = arrays["pairs"].view(-1, 2)[lo:hi].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(a, V, off, adj, eid, mine, d_te, ref_len)
        if not a.no_cpu_baseline and world == 1 and a.workload in CHEAPEST:
            mine = arrays["pairs"].view(-1, 2)[lo:hi].cpu().numpy()
            ns = len(mine) if a.workload == "forest_cheapest" else min(len(mine), 128)  # a Dijkstra on the knows graph is ~0.1 s
            out["cpu_baseline"] = cpu_baseline_cheapest(V, off, adj, eid, w, mine[:ns], d_val[:ns], d_ok[:ns])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(a, V, off, adj, eid, mine, d_te, ref_len):
    """The oracle's literal restatement of IterativeLengthFunction (reference memory layout and loops), driven in
    2048-row chunks like DuckDB drives the UDF.  Two bounded samples of this rank's pairs: one thread on the first
    `cpu_sample` pairs, and one thread per chunk (up to the host's cores) on as many chunks as there are cores.
    Checker + baseline only — never on the product path."""
    from oracle.pgq_oracle import OracleCSR
    cores = os.cpu_count() or 1
    ora = OracleCSR.adopt(V, off, adj, eid)
    ns1 = min(a.cpu_sample or (8192 if a.workload == "snb_sf100" else 1024), len(mine))
    t0 = time.perf_counter()
    ln, ok = ora.baseline_run("iterativelength", V, mine[:ns1, 0], mine[:ns1, 1], nthreads=1)
    dt1 = time.perf_counter() - t0
    gpu_len = ref_len[:ns1].cpu().numpy()
    agree = bool(((gpu_len >= 0) == ok).all() and (gpu_len[ok] == ln[ok]).all())
    te1 = float(d_te[:ns1].sum().item())
    # one worker per 2048-row chunk: the most threads DuckDB's chunking can use on these rows
    nchunks_all = max(1, (len(mine) + 2047) // 2048)
    threads = max(1, min(nchunks_all, cores))
    nsm = min(len(mine), threads * 2048)
    t0 = time.perf_counter()
    lnm, okm = ora.baseline_run("iterativelength", V, mine[:nsm, 0], mine[:nsm, 1], nthreads=threads)
    dtm = time.perf_counter() - t0
    gpu_m = ref_len[:nsm].cpu().numpy()
    agree_m = bool(((gpu_m >= 0) == okm).all() and (gpu_m[okm] == lnm[okm]).all())
    tem = float(d_te[:nsm].sum().item())
    return {"value": tem / dtm / 1e6, "unit": "MTEPS", "cores": threads, "kind": "port",
            "sample": "first %d pairs of rank 0's shard in 2048-row chunks, one thread per chunk (%d threads), literal "
                      "512-lane restatement (oracle/pgq_oracle.cpp), %.1f s; results equal the GPU's: %s" % (
                          nsm, threads, dtm, agree_m),
            "pairs_per_s": nsm / dtm, "host_cores_available": cores,
            "single_thread": {"value": te1 / dt1 / 1e6, "cores": 1, "pairs_per_s": ns1 / dt1,
                              "sample": "first %d pairs, %.1f s; results equal the GPU's: %s" % (ns1, dt1, agree)}}


def cpu_baseline_cheapest(V, off, adj, eid, w, mine, d_val, d_ok):
    """Per-pair Dijkstra of the oracle (lean restatement: same distances as the reference's batched Bellman-Ford, which
    needs 8 KiB per vertex per call and does not fit a 2^24-vertex graph) on this rank's pairs, one thread; every value
    compared with the GPU's bit for bit."""
    from oracle.pgq_oracle import OracleCSR
    ora = OracleCSR.adopt(V, off, adj, eid, w)
    t0 = time.perf_counter()
    want, wok = ora.lean_cheapest_path_length(V, mine[:, 0], mine[:, 1])
    dt = time.perf_counter() - t0
    ok = d_ok.cpu().numpy().astype(bool)
    got = d_val.cpu().numpy()
    if want.dtype.kind == "f":
        got = got.view(np.float64)
    agree = bool((ok == wok).all() and (got[ok] == want[wok]).all())
    return {"value": len(mine) / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "all %d pairs, per-pair Dijkstra (oracle/pgq_oracle.cpp lean restatement), %.1f s; results equal the "
                      "GPU's: %s" % (len(mine), dt, agree)}


if __name__ == "__main__":
    main()
