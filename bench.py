#!/usr/bin/env python3
"""bench.py — MS-BFS hot path of DuckPGQ on MI355X (contract: one JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over this rank's batch of (src,dst) pairs, inputs resident in HBM:
pgq_iterativelength_bulk_device (lane assignment, MS-BFS levels, per-pair hop counts), followed for N > 1 by the
RCCL all_gather of the per-pair lengths (the only inter-GPU traffic; the CSR is replicated).

Default workload (BASELINE.json metric "MS-BFS MTEPS + src-dst pairs/sec, SNB SF100, 1/2/4/8 GPU"): synthetic
LDBC-SNB-SF100-shaped Person-knows-Person graph (V=448,626, 39.88 M symmetric CSR entries), iterativelength,
8192 random pairs per GPU == configs[3] (65,536 pairs over 8 GPUs) cut to the per-GPU shard, weak scaling.
Other BASELINE configs: --workload rmat22 (configs[1]), snb_paths (configs[2]), forest_cheapest (configs[4]).

value   = MTEPS: traversed edges / second / 1e6, summed over ranks.  Traversed edges of a pair = out-degrees of
          all vertices its own level-synchronous BFS expands up to the level that reaches dst (all levels if
          unreachable) — a pure function of (graph, src, dst), counted once on the GPU outside the timed region
          (pgq_traversed_edges_bulk_device) and pinned against the CPU oracle in tests/.
roofline: dominant kernel class by HIP-event time inside the timed region (events recorded on the library's own
          stream around every launch); achieved = algorithmic bytes / event time (DESIGN.md has the formulas).
cpu_baseline: the literal restatement of the reference UDF (oracle/, 512-lane bitsets, 2048-row chunks) timed on
          this box's host cores on a bounded sample of the same pairs, same MTEPS definition.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling is reported beside it


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="snb_sf100", choices=["snb_sf100", "rmat22", "snb_paths", "forest_cheapest"])
    ap.add_argument("--pairs-per-gpu", type=int, default=0)
    ap.add_argument("--scale", type=int, default=0, help="override graph scale (rmat scale / forest log2 V); tests")
    ap.add_argument("--snb-vertices", type=int, default=448626)
    ap.add_argument("--snb-friendships", type=int, default=19_940_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs timed on the CPU (0 = 8192 snb / 2048 rmat)")
    ap.add_argument("--cpu-threads", type=int, default=1)
    ap.add_argument("--backend", default="nccl")
    return ap.parse_args()


def build_graph(a):
    from duckpgq_extension_amd import graphgen
    t0 = time.time()
    w = None
    if a.workload in ("snb_sf100", "snb_paths"):
        V, s, d = graphgen.snb_knows_like(a.snb_vertices, a.snb_friendships, seed=100)
        name = "snb_sf100_knows(V=%d)" % V
    elif a.workload == "rmat22":
        V, s, d = graphgen.rmat(a.scale or 22, seed=22)
        name = "rmat%d_ef16" % (a.scale or 22)
    else:
        V, s, d = graphgen.reply_forest(1 << (a.scale or 24), seed=5)
        w = np.random.default_rng(5).integers(1, 1000, len(s))
        name = "reply_forest(V=2^%d,int64 w)" % (a.scale or 24)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    if w is not None:
        w = w[eid]
    return name, V, off, adj, eid, w, time.time() - t0


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    import duckpgq_extension_amd as pgq

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=a.backend, rank=rank, world_size=world)
    n_gpus = world
    use_cuda = torch.cuda.is_available()
    if not use_cuda:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local % torch.cuda.device_count())
    dev = torch.device("cuda", local % torch.cuda.device_count())
    pgq.load_hip().pgq_init(local % torch.cuda.device_count())

    pairs_per_gpu = a.pairs_per_gpu or {"snb_sf100": 8192, "rmat22": 1024, "snb_paths": 4096,
                                        "forest_cheapest": 4096}[a.workload]
    # ---- graph: rank 0 builds it, the others receive it over RCCL (CSR replicated on every GPU) ----------------
    from duckpgq_extension_amd import sharding
    arrays = None
    name, gen_s = "", 0.0
    if rank == 0:
        name, V, off, adj, eid, w, gen_s = build_graph(a)
        arrays = {"off": torch.from_numpy(off), "adj": torch.from_numpy(adj), "eid": torch.from_numpy(eid)}
        if w is not None:
            arrays["w"] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.int64))
    arrays = sharding.broadcast_csr(arrays, dev)
    t_off, t_adj, t_eid, t_w = arrays["off"], arrays["adj"], arrays["eid"], arrays.get("w")
    has_w = t_w is not None
    V, E = t_off.numel() - 1, t_adj.numel()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    csr = pgq.DeviceCSR.from_device_ptrs(V, t_off.data_ptr(), t_adj.data_ptr(), t_eid.data_ptr(),
                                         t_w.data_ptr() if has_w else 0, 1 if has_w else 0)
    upload_s = time.perf_counter() - t0

    # ---- pairs: one global list, contiguous shard per rank -----------------------------------------------------
    seed = {"snb_sf100": 4, "rmat22": 2, "snb_paths": 3, "forest_cheapest": 5}[a.workload]
    total_pairs = pairs_per_gpu * world
    allp = np.random.default_rng(seed).integers(0, V, size=(total_pairs, 2))
    if a.workload == "forest_cheapest":  # destinations that are reachable at all: ancestors are rare, use edges' heads
        pass
    lo, hi = sharding.shard_bounds(total_pairs, world, rank)
    mine = allp[lo:hi]
    n = len(mine)
    d_src = torch.from_numpy(np.ascontiguousarray(mine[:, 0])).to(dev)
    d_dst = torch.from_numpy(np.ascontiguousarray(mine[:, 1])).to(dev)
    d_len = torch.empty(n, dtype=torch.int64, device=dev)
    d_te = torch.zeros(n, dtype=torch.int64, device=dev)

    child_cap = n * 64
    d_off = d_child = d_val = d_ok = None
    if a.workload == "snb_paths":
        d_off = torch.zeros(n, dtype=torch.int64, device=dev)
        d_child = torch.empty(child_cap, dtype=torch.int64, device=dev)
    if a.workload == "forest_cheapest":
        d_val = torch.zeros(n, dtype=torch.int64, device=dev)
        d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)

    def step():
        if a.workload == "snb_paths":
            rc, used = csr.shortestpath_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr(),
                                                 d_off.data_ptr(), d_child.data_ptr(), child_cap)
            assert rc == 0, pgq.load_hip().pgq_last_error()
        elif a.workload == "forest_cheapest":
            csr.cheapest_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_val.data_ptr(), d_ok.data_ptr())
        else:
            csr.iterativelength_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr())
        if world > 1:  # final RCCL gather of the per-pair results (xGMI)
            sharding.gather_rows(d_val if a.workload == "forest_cheapest" else d_len, total_pairs)

    # ---- work units (outside the timed region) -----------------------------------------------------------------
    if a.workload != "forest_cheapest":
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
    pgq.set_option("profile", 1)
    pgq.reset_stats()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    stats = pgq.get_stats()
    pgq.set_option("profile", 0)
    if a.workload != "forest_cheapest":
        assert bool((d_len == ref_len).all()), "results changed between passes"
    # Untimed extra pass with one batch in flight: in the timed region the batches of a call overlap on several HIP
    # streams, so a kernel's event duration there includes time it shared the GPU with other kernels.  The isolated
    # duration is reported beside it (roofline.isolated); `value` and roofline.achieved stay those of the timed region.
    n_streams = int(pgq.get_option("streams"))
    pgq.set_option("streams", 1)
    pgq.set_option("profile", 1)
    pgq.reset_stats()
    iso_steps = max(1, min(a.steps, 3))
    for _ in range(iso_steps):
        step()
    torch.cuda.synchronize()
    iso = pgq.get_stats()
    pgq.set_option("profile", 0)
    pgq.set_option("streams", n_streams)

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(te_local), float(stats["edges_scanned"])], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(el[0])
    te_total = float(tot[0])

    if rank == 0:
        kms, kb, kl = stats["kernel_ms"], stats["algo_bytes"], stats["launches"]
        dom = max(kms, key=lambda k: kms[k])
        ach = kb[dom] / 1e9 / (kms[dom] / 1e3) if kms[dom] > 0 else 0.0
        try:
            copy_gbps = pgq.copy_bandwidth_gbps(1 << 30, 5)
        except Exception:
            copy_gbps = None
        pairs_per_s = total_pairs * a.steps / elapsed
        if a.workload == "forest_cheapest":
            metric, unit = "cheapest_path_pairs_per_s", "pairs/s"
            value = pairs_per_s
        else:
            metric, unit = "msbfs_mteps", "MTEPS"
            value = te_total * a.steps / elapsed / 1e6
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_%s.json" % a.workload)
        if os.path.exists(pmc):  # written by tools/collect_pmc.py from separate rocprofv3 --pmc passes
            try:
                traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64" if a.workload != "forest_cheapest" else "int64", "data": "synthetic",
            "config": {"workload": "%s %s, %d pairs/GPU, CSR replicated" % (
                name, {"snb_sf100": "iterativelength", "rmat22": "iterativelength",
                       "snb_paths": "shortestpath+reconstruction", "forest_cheapest": "cheapest_path_length"}[a.workload],
                pairs_per_gpu), "V": V, "E": E, "pairs_total": total_pairs,
                "parallelism": "pairs sharded x%d, RCCL all_gather of lengths" % world if world > 1 else "1 GPU",
                "graph_gen_s": round(gen_s, 1), "csr_upload_ms": round(upload_s * 1e3, 2)},
            "pairs_per_s": pairs_per_s,
            "traversed_edges_per_step": te_total,
            # logical edges (value) vs the in/out-edges the kernels physically scanned in the timed region
            "mteps_physical": float(tot[1]) / elapsed / 1e6,
            "physical_edges_scanned_per_step": float(tot[1]) / a.steps,
            # the CSR dies at QueryEnd: one query = one upload (device-resident arrays here) + the searches
            "ms_per_step_incl_csr_upload": elapsed / a.steps * 1e3 + upload_s * 1e3,
            "levels_per_step": stats["levels"] / max(a.steps, 1),
            "push_pull_levels": [stats["push_levels"] // max(a.steps, 1), stats["pull_levels"] // max(a.steps, 1)],
            "kernel_ms_per_step": {k: round(v / a.steps, 4) for k, v in kms.items() if v},
            "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                         "launches": int(kl[dom]), "avg_launch_ms": kms[dom] / max(kl[dom], 1),
                         "algorithmic_bytes_per_launch": kb[dom] / max(kl[dom], 1),
                         "measured_copy_GBps": copy_gbps, "streams": n_streams,
                         "isolated": (lambda ms, by, ln: {
                             "achieved": by / 1e9 / (ms / 1e3), "frac": by / 1e9 / (ms / 1e3) / HBM_PEAK_GBPS,
                             "avg_launch_ms": ms / max(ln, 1), "streams": 1, "steps": iso_steps})(
                             iso["kernel_ms"][dom], iso["algo_bytes"][dom], iso["launches"][dom])
                         if iso["kernel_ms"].get(dom, 0) > 0 else None},
            # every kernel class of the timed region: event ms per step, algorithmic GB/s, launches per step
            "roofline_by_kernel": {k: {"ms_per_step": round(kms[k] / a.steps, 4),
                                       "GBps": round(kb[k] / 1e9 / (kms[k] / 1e3), 1) if kb[k] > 0 else None,
                                       "launches_per_step": kl[k] / a.steps}
                                   for k in kms if kms[k] > 0},
            "deferred_pairs_per_step": stats["deferred_pairs"] / max(a.steps, 1),
        }
        if not a.no_cpu_baseline and world == 1 and a.workload in ("snb_sf100", "rmat22"):  # rank 0, N=1 only
            out["cpu_baseline"] = cpu_baseline(a, V, off, adj, eid, mine, d_te, ref_len)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(a, V, off, adj, eid, mine, d_te, ref_len):
    """The oracle's literal restatement of IterativeLengthFunction (reference memory layout and loops), driven in
    2048-row chunks, on a bounded sample of this rank's pairs.  Checker + baseline only — never on the product path."""
    from oracle.pgq_oracle import OracleCSR
    ns = min(a.cpu_sample or (8192 if a.workload == "snb_sf100" else 2048), len(mine))
    ora = OracleCSR.adopt(V, off, adj, eid)
    t0 = time.perf_counter()
    ln, ok = ora.baseline_run("iterativelength", V, mine[:ns, 0], mine[:ns, 1], nthreads=a.cpu_threads)
    dt = time.perf_counter() - t0
    # the same sample with one worker per 2048-row chunk: the most threads DuckDB's chunking could use on it
    nchunks = max(1, (ns + 2047) // 2048)
    t0 = time.perf_counter()
    ora.baseline_run("iterativelength", V, mine[:ns, 0], mine[:ns, 1], nthreads=nchunks)
    dt_mt = time.perf_counter() - t0
    gpu_len = ref_len[:ns].cpu().numpy()
    agree = bool(((gpu_len >= 0) == ok).all() and (gpu_len[ok] == ln[ok]).all())
    te = float(d_te[:ns].sum().item())
    return {"value": te / dt / 1e6, "unit": "MTEPS", "cores": a.cpu_threads, "kind": "port",
            "sample": "first %d pairs of rank 0's shard, literal 512-lane restatement (oracle/pgq_oracle.cpp), "
                      "%.1f s; results equal the GPU's: %s" % (ns, dt, agree),
            "pairs_per_s": ns / dt, "host_cores_available": os.cpu_count(),
            "one_thread_per_chunk": {"threads": nchunks, "value": te / dt_mt / 1e6, "pairs_per_s": ns / dt_mt}}


if __name__ == "__main__":
    main()
