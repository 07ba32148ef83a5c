#!/usr/bin/env python3
"""Aggregates the rocprofv3 --pmc CSVs written by tools/collect_pmc.sh into per-kernel averages (per launch) and
writes profiles/pmc_<workload>.json (bench.py reads hbm_bytes_per_launch from it for roofline.traffic).
FETCH_SIZE/WRITE_SIZE are in KiB-ish units of 1024 B... rocprofv3 reports kilobytes; on gfx950 FETCH_SIZE under-counts
wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM): the 2x correction is applied to reads."""
import collections
import csv
import glob
import json
import os
import sys



def summarise(csv_paths):
    """Per-kernel averages per launch, class aliases and the pre-pass chain's traffic per step from rocprofv3 counter CSVs."""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for path in csv_paths:
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pgq::", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    return _summarise(agg, launches)


def _summarise(agg, launches):
    out = {}
    for k, cs in agg.items():
        d = {c: v / max(len(launches[(k, c)]), 1) for c, v in cs.items()}
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0
        d["launches_profiled"] = max(len(v) for (kk, c), v in launches.items() if kk == k)
        out[k] = d
    # bench.py looks kernels up by class name ("pull_sparse"): add aliases.  A class alias carries the launch-weighted average
    # of the byte counters over the class's kernels plus the name of the kernel that moves the most bytes per step.  Round 4:
    # k_meet3, the bit-map kernels and k_bibfs are classes of their own ("meet", "meet4", "bibfs"), so a class average no
    # longer mixes a 900-MB launch with a 0.1-MB one, and the pre-pass chain's traffic is stated PER STEP (sum over its kernels
    # of bytes x launches, divided by k_meet3's launches = the steps profiled).
    for cls, prefixes in (("pull_sparse", ("k_pull_lanes<", "k_pull_sparse<")), ("pull", ("k_pull<",)), ("push", ("k_push<",)),
                          ("pull_hub", ("k_pull_hub<",)), ("relax", ("k_relax<",)), ("meet", ("k_meet3",)),
                          ("meet4", ("k_meet4d", "k_meet4<")), ("bibfs", ("k_bibfs",)), ("ball", ("k_src_ball",))):
        cands = [k for k in out if k.startswith(prefixes)]
        if cands:
            best = max(cands, key=lambda k: out[k].get("hbm_bytes_per_launch", 0) * out[k]["launches_profiled"])
            launches_cls = sum(out[k]["launches_profiled"] for k in cands)
            alias = dict(out[best], kernel=best, kernels=sorted(cands), launches_profiled=launches_cls)
            for c in ("FETCH_SIZE", "WRITE_SIZE", "hbm_bytes_per_launch"):
                alias[c] = sum(out[k].get(c, 0.0) * out[k]["launches_profiled"] for k in cands) / max(launches_cls, 1)
            out[cls] = alias
    if "meet" in out:
        # steps = launches of the k_meet3 variant that moves the most bytes (round 5: the 1024-row calibration call of a fresh
        # CSR launches the small-call variant once; counted as a step it made the chain look 12 % lighter than its own first kernel)
        m3 = [k for k in out if k.startswith(("k_meet3", "k_src_ball"))]  # round 6: or of k_src_ball, when IT takes the calls (k_meet3 then returns at once)
        top = max(m3, key=lambda k: out[k].get("hbm_bytes_per_launch", 0) * out[k]["launches_profiled"]) if m3 else None
        steps = max(out[top]["launches_profiled"] if top else out["meet"]["launches_profiled"], 1)
        chain = [k for k in out if k.startswith(("k_meet3", "k_meet4", "k_bibfs", "k_src_ball", "k_ball_segments"))]
        out["prepass_chain"] = {"steps_profiled": steps, "kernels": sorted(chain),
                                "hbm_bytes_per_step": sum(out[k].get("hbm_bytes_per_launch", 0.0) * out[k]["launches_profiled"] for k in chain) / steps}
    # round 5: the lane-batched search's launch chain per step (bench.py: roofline.traffic of the msbfs legs) — every kernel
    # of a batch from the lane assignment to the result scatter; steps profiled = launches of k_init_batch (one per batch)
    init = [k for k in out if k.startswith("k_init_batch")]
    if init:
        steps = max(sum(out[k]["launches_profiled"] for k in init), 1)
        names = ("k_prep_zero", "k_mark_sources", "k_compact_sources", "k_pair_rows", "k_pair_keys", "k_gather_sorted", "k_batch_bounds",
                 "k_batch_reset", "k_init_batch", "k_level_reset", "k_push<", "k_clear_items", "k_queue_from_dense", "k_detect<",
                 "k_compact_lanes", "k_pull_lanes<", "k_pull<", "k_pull_hub", "k_probe<", "k_probe2<", "k_scatter_results",
                 "k_clean_by_nz", "k_open_merge", "k_compact_frontier", "k_pull_sparse<")
        chain = [k for k in out if k.startswith(names)]
        out["chain"] = {"steps_profiled": steps, "kernels": sorted(chain),
                        "hbm_bytes_per_step": sum(out[k].get("hbm_bytes_per_launch", 0.0) * out[k]["launches_profiled"] for k in chain) / steps}
    return out


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "snb_sf100"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = summarise(glob.glob(os.path.join(root, "gpurun_out", "prof", wl + "_[A-Z]", "*counter_collection.csv")))
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(root, "profiles", "pmc_%s.json" % wl), "w"), indent=1, sort_keys=True)
    for k in sorted(out, key=lambda k: -out[k].get("hbm_bytes_per_launch", 0)):
        if k.startswith("k_"):
            print(k, {c: "%.4g" % v for c, v in out[k].items() if isinstance(v, float)})
