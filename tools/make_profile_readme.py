#!/usr/bin/env python3
"""Writes profiles/README.md from the artefacts of one measurement pass (tools/measure_pass.sh on the GPU box, copied to
profiles/rNN): bench JSON lines, rocprofv3 kernel stats, PMC summaries (tools/pmc_summary.py), chunk latencies and the
memory micro-benchmarks.  usage: python tools/make_profile_readme.py [r02]"""
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
rdir = os.path.join(root, "profiles", rnd)


def load(name):
    p = os.path.join(rdir, name)
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        return None
    try:
        return json.load(open(p))
    except ValueError:
        return None


CLASS_OF = {"k_meet3": "meet", "k_meet4d": "meet4", "k_bibfs": "bibfs", "k_pull_lanes": "pull_sparse"}


def fmt(x, spec="{:,.0f}"):
    return spec.format(x) if x is not None else "—"


L = ["# profiles — round %s measurements (MI355X, one GPU per gpurun box)" % rnd.lstrip("r0"), "",
     "Produced by committed tooling only: `tools/measure_pass_r04.sh` (round 3: `tools/measure_pass.sh`) on the GPU box runs `bench.py` per workload (one JSON "
     "line each), `rocprofv3 --kernel-trace --stats` of the default bench command without its second leg (`--no-legs`: "
     "every `k_meet3` / `k_meet4d` call is a 65,536-row one) and of the cross-product workload (`--workload snb_cross`), "
     "separate `--pmc` passes (round 4: of the default workload; the cross-product file is round 3's) summarised by `tools/pmc_summary.py` into `profiles/pmc_<workload>.json` ("
     "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM), `tools/chunk_latency.py`; the `tools/membench` figures quoted in DESIGN.md are round 3's (`profiles/r03/membench_*`).  "
     "Regenerate this file with `python tools/make_profile_readme.py %s`.  `profiles/r01/` … `profiles/r03/` are the "
     "previous rounds." % rnd,
     ""]
names = [("snb_sf100", "C4 shard: SF100-shaped knows, iterativelength, 65,536 random pairs (default bench; every row through the pre-pass)"),
         ("snb_sf100_8192", "same graph, 8192 random pairs"),
         ("snb_sf100_2048", "same graph, 2048 random pairs (one DuckDB chunk's worth, device arrays)"),
         ("snb_sf100_8192_msbfs_only", "same, 8192 random pairs, `PGQ_MEET=0` (lane-batched MS-BFS only)"),
         ("snb_cross", "same graph, cross product 2048 sources x 1024 destinations = 2.1 M rows (lane-batched MS-BFS; the `msbfs_cross` leg)"),
         ("snb_cross_2048x32", "same graph, cross product 2048 sources x 32 destinations = 65,536 rows"),
         ("snb_cross_allv", "same graph, 32 sources x every vertex = 14.4 M rows"),
         ("rmat22", "C2: R-MAT scale 22, iterativelength, 1024 pairs"),
         ("snb_paths", "C3: SF100-shaped knows, shortestpath + reconstruction, 4096 pairs"),
         ("forest_cheapest", "C5: reply forest V=2^24, int64 weights, cheapest_path_length, 4096 reachable pairs"),
         ("forest_cheapest_double", "C5, double weights"),
         ("forest_cheapest_2_28", "C5 at the named scale: reply forest V=2^28 (268 M vertices, 215 M edges), int64 weights, 4096 pairs"),
         ("snb_cheapest_4096", "general graph: weighted knows graph (int64 weights 1..999), cheapest_path_length, 4096 pairs (batched relaxation, light edges first; 3 batches side by side)"),
         ("snb_cheapest_4096_double", "same, double weights"),
         ("snb_cheapest_4096_streams6", "same, int64, 6 batches side by side (`relax_streams=6`, the default since; the kernel columns of this line come from a pass in which the 6 batches overlapped — only ms/step and pairs/s count)")]
L += ["## bench.py, 1 GPU (10 steps, 2 warm-up; timed region runs unprofiled, the roofline columns come from an untimed "
      "pass with one batch in flight and HIP events around every kernel)", "",
      "| workload | ms/step | pairs/s | MTEPS physical (adjacency entries scanned) | rows answered by the pre-pass | dominant kernel | launches/step | "
      "avg launch ms | algorithmic GB/s | frac of 8 TB/s | whole step GB/s (frac) | CPU baseline |",
      "|---|---|---|---|---|---|---|---|---|---|---|---|"]
for w, title in names:
    j = load("bench_%s.json" % w)
    if not j:
        continue
    r = j.get("roofline") or {}
    step = r.get("step") or {}
    cpu = j.get("cpu_baseline")
    st = (cpu or {}).get("single_thread")
    cpu_s = "—"
    if cpu:
        cpu_s = "%s %s on %d threads" % (fmt(cpu["value"]), cpu["unit"], cpu["cores"])
        if st:
            cpu_s += "; %s on 1" % fmt(st["value"])
    L.append("| %s | %.3f | %s | %s | %s | `%s` | %s | %s | %s | %s | %s | %s |" % (
        title, j["ms_per_step"], fmt(j.get("pairs_per_s")), fmt(j.get("mteps_physical")) if j.get("mteps_physical") else "—",
        fmt(j.get("rows_answered_by_prepass_per_step")), r.get("kernel", "—"),
        fmt((j.get("roofline_by_kernel") or {}).get(CLASS_OF.get(r.get("kernel", ""), r.get("kernel", "").replace("k_", "")), {}).get("launches_per_step"), "{:.1f}"),
        fmt(r.get("avg_launch_ms"), "{:.3f}"), fmt(r.get("achieved")), fmt(r.get("frac"), "{:.3f}"),
        "%s (%s)" % (fmt(step.get("GBps")), fmt(step.get("frac"), "{:.3f}")) if step else "—", cpu_s))
L += ["", "Kernel classes of the untimed one-batch-in-flight pass (ms per step, algorithmic GB/s where the class has a "
      "byte model):", ""]
for w, _ in names:
    j = load("bench_%s.json" % w)
    if j and j.get("roofline_by_kernel"):
        fe = (j.get("roofline") or {}).get("frontier_expansion")
        pc = (j.get("roofline") or {}).get("prepass_chain")
        L.append("* **%s**: " % w + ", ".join(
            "%s %.3f ms%s" % (k, v["ms_per_step"], (" (%.0f GB/s)" % v["GBps"]) if v.get("GBps") else "")
            for k, v in j["roofline_by_kernel"].items()) +
            ("; frontier expansion (push + pull + pull_sparse) %.3f ms at %.0f GB/s = %.3f of peak" % (
                fe["ms_per_step"], fe["GBps"], fe["frac"]) if fe else "") +
            ("; pre-pass chain (%s) %.3f ms at %.0f GB/s = %.3f of peak" % (
                " + ".join(pc["classes"]), pc["ms_per_step"], pc["GBps"], pc["frac"]) if pc and pc["GBps"] > 1 else ""))
L += ["", "## rocprofv3 --kernel-trace --stats (top kernels)", ""]
for p in sorted(glob.glob(os.path.join(rdir, "*kernel_stats.csv"))):
    L += ["`profiles/%s/%s`" % (rnd, os.path.basename(p)), "", "| kernel | calls | avg µs | % |", "|---|---|---|---|"]
    for r in list(csv.DictReader(open(p)))[:10]:
        L.append("| `%s` | %s | %.1f | %s |" % (r["Name"].split("(")[0].replace("void ", "").replace("pgq::", ""),
                                                 r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    L.append("")
L += ["## PMC summaries (per launch, averaged over the profiled launches)", ""]
keep = ("FETCH_SIZE", "WRITE_SIZE", "hbm_bytes_per_launch", "launches_profiled", "TCC_HIT_sum", "TCC_MISS_sum",
        "TA_TA_BUSY_sum", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD")
for p in sorted(glob.glob(os.path.join(root, "profiles", "pmc_*.json"))):
    j = json.load(open(p))
    L += ["`profiles/%s`" % os.path.basename(p), "",
          "| kernel class | launches | HBM MB/launch (2×FETCH_SIZE KB + WRITE_SIZE KB) | L2 hit rate | "
          "VALU insts / VMEM reads | wave cycles waiting on instructions |", "|---|---|---|---|---|---|"]
    for k, v in j.items():
        if k.startswith("__") or not isinstance(v, dict) or not v.get("hbm_bytes_per_launch"):
            continue
        hit, miss = v.get("TCC_HIT_sum"), v.get("TCC_MISS_sum")
        L.append("| `%s` | %s | %.1f | %s | %s | %s |" % (
            k, fmt(v.get("launches_profiled")), v["hbm_bytes_per_launch"] / 1e6,
            "%.2f" % (hit / (hit + miss)) if hit is not None and miss is not None and hit + miss > 0 else "—",
            "%.1f" % (v["SQ_INSTS_VALU"] / v["SQ_INSTS_VMEM_RD"]) if v.get("SQ_INSTS_VMEM_RD") else "—",
            "%.2f" % (v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]) if v.get("SQ_WAVE_CYCLES") else "—"))
    L.append("")
cl = load("chunk_latency.json")
if cl:
    L += ["## C-ABI call latencies (`tools/chunk_latency.py`, SF100-shaped graph, ms)", "",
          "| call | ms |", "|---|---|"] + ["| %s | %.3f |" % (k.replace("_ms", ""), v) for k, v in cl.items()] + [""]
mc = os.path.join(rdir, "membench_copy.jsonl")
if os.path.exists(mc):
    rows = [json.loads(x) for x in open(mc) if x.startswith("{")]
    if rows:
        best = max(rows, key=lambda r: r["GBps_read_plus_write"])
        L += ["## HBM ceiling of the box (`tools/membench copy`, 2 GiB, read + written bytes / time)", "",
              "best shape: `%s` grid %d × %d threads: **%.0f GB/s**; range over %d shapes %.0f – %.0f GB/s.  The guide's "
              "8 TB/s is the denominator of every `frac` above; this is what a plain copy reaches on the same box." % (
                  best["kernel"], best["grid"], best["block"], best["GBps_read_plus_write"], len(rows),
                  min(r["GBps_read_plus_write"] for r in rows), best["GBps_read_plus_write"]), ""]
ms_ = os.path.join(rdir, "membench_segments.jsonl")
if os.path.exists(ms_):
    rows = [json.loads(x) for x in open(ms_) if x.startswith("{")]
    if rows:
        L += ["## Ceiling for the list walks (`tools/membench segments`: whole segments at random 16-byte-aligned starts, "
              "16 bytes per lane, four requests per wavefront in flight)", "",
              "| table | segment | waves/SIMD | GB/s |", "|---|---|---|---|"]
        for r in rows:
            L.append("| %d MB | %d B | %d | %.0f |" % (r["table_MB"], r["segment_bytes"], r["waves_per_simd"], r["GBps"]))
        L += ["", "`k_meet3` / `k_meet4` read adjacency lists of a few hundred entries (0.6 KB on average on the SF100-shaped "
              "graph) picked by a pair's one-hop list out of the 160 MB adjacency: the 1 KB rows are the ceiling for that "
              "pattern on this box.", ""]
mg = os.path.join(rdir, "membench_gather.jsonl")
if os.path.exists(mg):
    rows = [json.loads(x) for x in open(mg) if x.startswith("{")]
    fetch = {}
    for p in glob.glob(os.path.join(rdir, "membench_gather", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == "FETCH_SIZE" and "k_gather" in r["Kernel_Name"]:
                fetch.setdefault(16 if "<16>" in r["Kernel_Name"] else 32, []).append(float(r["Counter_Value"]))
    if rows:
        L += ["## FETCH_SIZE against a gather of known size (`tools/membench gather` under `--pmc FETCH_SIZE`)", "",
              "64 Mi random record reads (16 or 32 bytes, records aligned to their size); the index list (4 B per read) "
              "streams.  `raw` is the counter as rocprofv3 reports it (KB); the guide's correction doubles it.", "",
              "| kernel | table | G reads/s | raw FETCH_SIZE (MB) | raw bytes per read (index bytes removed at 2x) | 2x (MB) |",
              "|---|---|---|---|---|---|"]
        for r in rows:
            rec = int(r["kernel"].replace("gather", ""))
            vals = sorted(fetch.get(rec) or [])
            raw = (vals[0] if r["table_MB"] <= 4 else vals[-1]) if vals else None
            per = (raw * 1024 - r["index_bytes"] / 2) / r["reads"] if raw else None
            L.append("| %s | %d MB | %.1f | %s | %s | %s |" % (
                r["kernel"], r["table_MB"], r["Greads_per_s"], fmt(raw * 1024 / 1e6 if raw else None),
                fmt(per, "{:.1f}"), fmt(raw * 2 * 1024 / 1e6 if raw else None)))
        L += ["", "Reading: a random read of a 1 GB table costs one 64-byte request in the raw counter (the streamed index "
              "list is counted at half its size, as the guide says of wide coalesced reads).  If the fabric moves 64 bytes "
              "per such request, the doubled figure over-states gather traffic by up to 2x; if it moves 128, the doubled "
              "figure is exact.  `roofline.traffic` and the PMC tables above use the guide's 2x throughout, so for the "
              "gather-heavy kernels (`k_pull_lanes`, `k_pull`) they are upper bounds; for the list walks of `k_meet3` "
              "(1 KB per wavefront request) the 2x figure is the calibrated one.", ""]
open(os.path.join(root, "profiles", "README.md"), "w").write("\n".join(L) + "\n")
print("wrote profiles/README.md (%d lines)" % len(L))
