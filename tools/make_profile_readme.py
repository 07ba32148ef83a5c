#!/usr/bin/env python3
"""Writes profiles/README.md from the artefacts under profiles/r01 (bench JSON lines, rocprofv3 kernel stats,
PMC summaries).  Run after copying a measurement pass back from the GPU box."""
import csv
import json
import os

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r01 = os.path.join(root, "profiles", "r01")
L = ["# profiles — round 1 measurements (MI355X, one GPU per gpurun box)", "",
     "Everything here is produced by committed tooling: `bench.py` (JSON lines), "
     "`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py …` (kernel stats), "
     "`tools/collect_pmc.sh` + `tools/pmc_summary.py` (separate `--pmc` passes → `pmc_<workload>.json`, which "
     "`bench.py` reads for `roofline.traffic`; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM), "
     "`tools/chunk_latency.py`.  Regenerate this file with `python tools/make_profile_readme.py`.", ""]
names = {"snb_sf100": "C4 shard: SF100-shaped knows, iterativelength, 8192 pairs (default bench)",
         "rmat22": "C2: R-MAT scale 22, iterativelength, 1024 pairs",
         "snb_paths": "C3: SF100-shaped knows, shortestpath + reconstruction, 4096 pairs",
         "forest_cheapest": "C5: reply forest V=2^24, int64 weights, cheapest_path_length, 4096 pairs"}
L += ["## bench.py, 1 GPU (10 steps, 2 warm-up)", "",
      "| workload | ms/step | pairs/s | MTEPS | dominant kernel | achieved GB/s (algorithmic, timed region) | "
      "frac of 8 TB/s | same kernel, one batch in flight: GB/s (frac) | "
      "PMC traffic / launch | CPU baseline (1 core, literal restatement) |", "|---|---|---|---|---|---|---|---|---|---|"]
for w, title in names.items():
    p = os.path.join(r01, "bench_%s.json" % w)
    if not os.path.exists(p):
        continue
    j = json.load(open(p))
    r = j["roofline"]
    pmc_path = os.path.join(root, "profiles", "pmc_%s.json" % w)
    if os.path.exists(pmc_path):  # the PMC passes run after the bench line was written: take the fresh figure
        r["traffic"] = json.load(open(pmc_path)).get(r["kernel"].replace("k_", ""), {}).get("hbm_bytes_per_launch")
    cpu = j.get("cpu_baseline")
    iso = r.get("isolated")
    L.append("| %s | %.2f | %s | %s | `%s` | %.0f | %.3f | %s | %s | %s |" % (
        title, j["ms_per_step"], "{:,.0f}".format(j["pairs_per_s"]),
        "{:,.0f}".format(j["value"]) if j["unit"] == "MTEPS" else "—", r["kernel"], r["achieved"], r["frac"],
        "%.0f (%.3f)" % (iso["achieved"], iso["frac"]) if iso else "—",
        "%.0f MB (algorithmic %.0f MB)" % (r["traffic"] / 1e6, r["algorithmic_bytes_per_launch"] / 1e6)
        if r.get("traffic") else "—",
        "%.0f MTEPS, %.0f pairs/s" % (cpu["value"], cpu["pairs_per_s"]) if cpu else "—"))
L += ["", "Per-kernel-class HIP-event time inside the timed region (ms per step; three batches overlap on three "
      "streams, so the classes sum to more than the wall time and a kernel's duration includes time it shared the GPU "
      "with other kernels; `roofline.isolated` in the JSON is the same kernel with one batch in flight) and "
      "algorithmic GB/s:", ""]
for w in names:
    p = os.path.join(r01, "bench_%s.json" % w)
    if os.path.exists(p):
        j = json.load(open(p))
        L.append("* **%s**: " % w + ", ".join(
            "%s %.2f ms%s" % (k, v["ms_per_step"], (" (%.0f GB/s)" % v["GBps"]) if v.get("GBps") else "")
            for k, v in j["roofline_by_kernel"].items()))
L += ["", "## rocprofv3 --kernel-trace --stats (same command, top kernels)", ""]
for w in ("snb_sf100", "rmat22"):
    p = os.path.join(r01, "%s_kernel_stats.csv" % w)
    if not os.path.exists(p):
        continue
    L += ["`profiles/r01/%s_kernel_stats.csv`" % w, "", "| kernel | calls | avg µs | % |", "|---|---|---|---|"]
    for r in list(csv.DictReader(open(p)))[:8]:
        L.append("| `%s` | %s | %.1f | %s |" % (r["Name"].split("(")[0].replace("void ", ""), r["Calls"],
                                                 float(r["AverageNs"]) / 1e3, r["Percentage"]))
    L.append("")
L += ["Reading the two together: `bench.py`'s class timer brackets `k_compact_frontier` + `k_pull_sparse` of the 40 "
      "launches inside the timed region (three batches overlapping: ≈ 0.98 ms per pair of kernels); rocprofv3 averages "
      "`k_pull_sparse` alone over all 64 launches of the process: 12 of them belong to the untimed one-batch-in-flight "
      "pass (≈ 0.44 ms each, `roofline.isolated` minus the 0.055 ms compaction), which leaves ≈ 0.83 ms for each of the "
      "52 overlapped ones (warm-up, timed region, traversed-edge accounting pass).  The event-bracketed figure is "
      "higher (0.98 − 0.055 = 0.93 ms) because HIP events also see the time a launch waits behind the other streams' "
      "kernels.", ""]
L += ["## PMC (per launch averages, `profiles/pmc_<workload>.json`)", ""]
for w in ("snb_sf100", "rmat22"):
    p = os.path.join(root, "profiles", "pmc_%s.json" % w)
    if not os.path.exists(p):
        continue
    pm = json.load(open(p))
    for cls in ("pull_sparse", "pull", "push"):
        if cls in pm:
            d = pm[cls]
            hit = d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1)
            L.append("* %s `%s`: HBM-side bytes %.0f MB, L2 hit %.0f %%, wave cycles waiting %.0f %%, "
                     "TCP pending-stall/TA-busy cycles %.2g / %.2g" % (
                         w, d.get("kernel", cls), d.get("hbm_bytes_per_launch", 0) / 1e6, 100 * hit,
                         100 * d.get("SQ_WAIT_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1),
                         d.get("TCP_PENDING_STALL_CYCLES_sum", 0), d.get("TA_TA_BUSY_sum", 0)))
p = os.path.join(r01, "chunk_latency.json")
if os.path.exists(p):
    c = json.load(open(p))
    L += ["", "## Chunk entry points (host buffers in/out, what one DuckDB DataChunk costs; SF100-shaped graph)", "",
          "| call | ms |", "|---|---|"]
    L += ["| %s | %.2f |" % (k, v) for k, v in c.items()]
L += ["", "## Optimisation history on the default workload (8192 pairs, ms per call, same graph)", "",
      "| step | ms | note |", "|---|---|---|",
      "| first correct version: top-down level 1, dense bottom-up after | 20.5 | dense `k_pull` at 5.3 TB/s algorithmic (66 % of spec peak) |",
      "| + non-empty-word masks, destination probe | 16.2 | last level disappears: pairs are answered one expansion early |",
      "| + straggler deferral | 12.3 | full-width levels that served ~1 % of the pairs are re-run narrow |",
      "| + packed frontier, frontier bit map in LDS, fused records | 10.8 | level 2: 0.95 → 0.67 ms |",
      "| + fused per-level reset kernel | 9.9 | fewer tiny launches |",
      "| + two batches in flight on two streams | 7.5 | hides the per-level host round trip |",
      "| + finer top-down items, contention-free packing | 6.5 | |",
      "| + owner index per in-edge (no binary search), 2048-lane batches (WD=32) | 5.5 | the sparse kernel is issue-bound: fewer instructions, better lane use |",
      "| + two-hop destination probe | 4.1 | distance-4 pairs answered from the level-2 frontier: no straggler pass |",
      "| + top-down level without the shared queue counter | 3.6 | ~10^4 serialised atomicAdds per launch removed |",
      "| + accumulate step of the sparse kernel: one word of all 4 chunks per trip, next adjacency prefetched | 3.0 | "
      "one L2 round trip per trip instead of per word per chunk (kernel 0.77 -> 0.53 ms alone) |",
      "| + long-tail words spread over the wavefront through an LDS queue, 3 batches in flight | 2.8 | the fullest of "
      "256 entries holds 10.7 words, the average 1.2; 96 VGPRs leave room for the other streams' kernels |",
      "| + three words inline in 32-byte frontier records, 128 entries in flight per wavefront | 2.6 | 86 % of the hot "
      "entries need no second fetch; 78 VGPRs |", "",
      "R-MAT-22 (1024 pairs): 117 ms (first version, one wavefront per vertex dealt round-robin: R-MAT's id/degree "
      "correlation left a few wavefronts with all hubs) -> 12.1 ms (edge-balanced work parts, dead-destination marking) "
      "-> 6.9 ms (no lanes for pairs that cannot have a path: 1024 pairs -> 220 lanes, WD=4) -> 3.5 ms on the sweep's "
      "pair set / 8.9 ms on bench.py's (two-hop probe, contention-free hub statistics).",
      "shortestpath on SF100 (4096 pairs, full [v,e,...] reconstruction): 8.1 -> 3.3 ms (sparse level 2, straggler "
      "deferral with path append).  cheapest_path_length on the 2^24-vertex reply forest (4096 pairs): 35.5 -> 4.5 ms "
      "(device-side rounds for small frontiers, no lanes for unreachable pairs).", ""]
open(os.path.join(root, "profiles", "README.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L[:30]))
