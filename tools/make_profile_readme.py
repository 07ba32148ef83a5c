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
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
rdir = os.path.join(root, "profiles", rnd)


def load(name):
    p = os.path.join(rdir, name)
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        return None
    try:
        return json.load(open(p))
    except ValueError:
        return None


CLASS_OF = {"k_meet3": "meet", "k_meet4d": "meet4", "k_bibfs": "bibfs", "k_pull_lanes": "pull_sparse", "k_src_ball": "ball"}


def fmt(x, spec="{:,.0f}"):
    return spec.format(x) if x is not None else "—"


L = ["# profiles — round %s measurements (MI355X, one GPU per gpurun box)" % rnd.lstrip("r0"), "",
     "Produced by committed tooling only: `tools/measure_pass_%s.sh` on the GPU box (PART=1: `bench.py` with the driver's own "
     "command — every BASELINE config and call shape is a leg of that one line — and the shapes beside it on their routes, "
     "`rocprofv3 --kernel-trace --stats` of the default workload without its legs, of the SF100 cross product as routed "
     "(`k_src_ball`) and forced through the lane batches (`PGQ_BALL=0`), and of the R-MAT-22 cross product, "
     "`tools/chunk_latency.py`; PART=2: separate `--pmc` passes of those four workloads, summarised by `tools/pmc_summary.py` "
     "into `profiles/pmc_<workload>.json` (FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM)).  `tools/h2dbench.hip` is the "
     "PCIe / host-copy yardstick of the upload path (`profiles/%s/h2dbench.txt`).  Regenerate this file with "
     "`python tools/make_profile_readme.py %s`.  `profiles/r01/` … are the previous rounds." % (rnd, rnd, rnd),
     ""]
default = load("bench_default.json")
rows = []  # (title, leg-like dict)
if default:
    top = dict(default)
    rows.append(("configs[3] on one GPU: SF100-shaped knows, iterativelength, 65,536 random pairs (the line's top level = leg `prepass`)", top))
    titles = {"msbfs_cross": "same graph, the binder's call shape: 2048 sources x 1024 destinations = 2.1 M rows grouped by source (leg `msbfs_cross`, as routed: the source-centric kernel `k_src_ball`)",
              "msbfs_cross_lanes": "the same rows forced through the lane-batched MS-BFS (leg `msbfs_cross_lanes`, `ball = 0`)",
              "msbfs_cross_shuffled": "the same rows in random order, as routed: sorted by source for `k_src_ball` (leg `msbfs_cross_shuffled`)",
              "msbfs_cross_rmat22": "R-MAT scale 22 in the binder's call shape: 2048 x 1024 rows (leg `msbfs_cross_rmat22`, as routed)",
              "snb_paths": "configs[2]: SF100-shaped knows, shortestpath + reconstruction, 4096 pairs (leg `snb_paths`)",
              "rmat22": "configs[1]: R-MAT scale 22, iterativelength, 1024 pairs (leg `rmat22`)",
              "forest_cheapest": "configs[4]: reply forest V = 2^24, int64 weights, cheapest_path_length, 4096 pairs (leg `forest_cheapest`)",
              "cheapest_general": "configs[4]'s operator on a general graph: weighted knows graph, 4096 pairs (leg `cheapest_general`, one step)"}
    for k in ("msbfs_cross", "msbfs_cross_lanes", "msbfs_cross_shuffled", "msbfs_cross_rmat22", "snb_paths", "rmat22", "forest_cheapest", "cheapest_general"):
        if k in (default.get("legs") or {}):
            rows.append((titles[k], default["legs"][k]))
for w, title in (("snb_sf100_8192", "SF100 graph, 8192 random pairs (configs[3]'s shard at 8 GPUs)"),
                 ("snb_sf100_2048", "SF100 graph, 2048 random pairs (one DuckDB chunk's worth, device arrays)"),
                 ("snb_cross_2048x32", "cross product 2048 sources x 32 destinations = 65,536 rows (the device's decision: pre-pass)"),
                 ("snb_cross_2048x128", "cross product 2048 sources x 128 destinations = 262,144 rows (source-centric kernel)"),
                 ("rmat22_cross_lanes", "R-MAT-22 cross product through the lane batches alone (`PGQ_MEET=0`: no source-centric kernel, no pre-pass)"),
                 ("snb_cross_shuffled", "the 2048 x 1024 cross product with its rows in random order (a hash join's output): sorted by source for `k_src_ball` (`ball_sort`)"),
                 ("snb_cross_shuffled_nosort", "... the same rows with `ball_sort = 0`: lane batches"),
                 ("rmat22", "configs[1] as a workload of its own (20 steps)"),
                 ("rmat22_cross", "R-MAT-22 cross product as a workload of its own (as routed)"),
                 ("snb_cheapest_512", "weighted knows graph, 512 pairs, one lane per source (shipped)"),
                 ("snb_cheapest_512_bidir", "... the same pairs searched from both ends (`relax_bidir = 1`, off as shipped)"),
                 ("snb_cross_allv", "32 sources x every vertex = 14.4 M rows"),
                 ("forest_cheapest_double", "configs[4], double weights"),
                 ("forest_cheapest_2_28", "configs[4] at the named scale: reply forest V = 2^28 (268 M vertices, 215 M edges), int64 weights")):
    j = load("bench_%s.json" % w)
    if j:
        rows.append((title, j))
L += ["## bench.py, 1 GPU (timed region runs unprofiled; the roofline columns come from an untimed pass with one batch in "
      "flight and HIP events around every launch; `chain` = all kernel classes of the leg: algorithmic bytes over the sum of "
      "their launch durations)", "",
      "| workload | ms/step | pairs/s | launch chain | chain GB/s | chain frac of 8 TB/s | chain traffic (PMC) / algorithmic | dominant kernel (frac) | "
      "whole step frac | CPU baseline |",
      "|---|---|---|---|---|---|---|---|---|---|"]
for title, j in rows:
    r = j.get("roofline") or {}
    step = r.get("step") or {}
    dom = r.get("dominant_kernel") or {}
    cpu = j.get("cpu_baseline")
    cpu_s = "—"
    if cpu and cpu.get("value") is not None:
        cpu_s = "%s %s on %s threads" % (fmt(cpu["value"]), cpu.get("unit", ""), cpu.get("cores"))
    # traffic: from the PMC files as committed (the bench line embeds what the file held when it ran), and only where the
    # counters were collected on this shape: the default workload and the 2048 x 1024 cross product
    tr = None
    pmc_of = {"prepass": ("pmc_snb_sf100.json", "prepass_chain"), "msbfs_cross": ("pmc_snb_cross_ball.json", "prepass_chain"),
              "msbfs_cross_lanes": ("pmc_snb_cross.json", "chain"), "msbfs_cross_rmat22": ("pmc_rmat22_cross.json", "prepass_chain")}
    key = "prepass" if j is rows[0][1] else next((k for k in ("msbfs_cross_rmat22", "msbfs_cross_lanes", "msbfs_cross") if "`%s`" % k in title), None)
    if key:
        try:
            tr = json.load(open(os.path.join(root, "profiles", pmc_of[key][0])))[pmc_of[key][1]]["hbm_bytes_per_step"]
        except Exception:
            tr = None
    ab = r.get("algorithmic_bytes_per_step")
    L.append("| %s | %.4f | %s | %s | %s | %s | %s | `%s` (%s) | %s | %s |" % (
        title, j["ms_per_step"], fmt(j.get("pairs_per_s")), " + ".join(r.get("classes") or []) or "—", fmt(r.get("achieved")),
        fmt(r.get("frac"), "{:.3f}"), ("%.0f MB / %.0f MB = %.2f x" % (tr / 1e6, ab / 1e6, tr / ab)) if tr and ab else "—",
        dom.get("kernel", "—"), fmt(dom.get("frac"), "{:.3f}"), fmt(step.get("frac"), "{:.3f}"), cpu_s))
L += ["", "Kernel classes of the untimed one-batch-in-flight pass (ms per step; algorithmic GB/s where the class has a byte model):", ""]
for title, j in rows:
    if j.get("roofline_by_kernel"):
        L.append("* **%s**: " % title.split(":")[0].split("(")[0].strip() + ", ".join(
            "%s %.3f ms%s" % (k, v["ms_per_step"], (" (%.0f GB/s)" % v["GBps"]) if v.get("GBps") else "")
            for k, v in j["roofline_by_kernel"].items()))
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
ct = os.path.join(rdir, "chunk_throughput.txt")
if os.path.exists(ct) and os.path.getsize(ct):
    try:
        tj = json.loads(open(ct).read().strip().splitlines()[-1])
        L += ["## Concurrent chunk calls (`tools/chunk_throughput.py` -> `tools/chunk_mt.cpp`: T native threads, one 2048-row "
              "`pgq_iterativelength` call per DataChunk each, one shared CSR)", "", "| shape, threads | rows/s | ms per chunk |", "|---|---|---|"]
        L += ["| %s | %s | %.4f |" % (k, fmt(v["rows_per_s"]), v["ms_per_chunk"]) for k, v in tj.items()] + [""]
    except ValueError:
        pass
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
