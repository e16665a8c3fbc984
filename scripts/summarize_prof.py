#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by scripts/profile_bench.sh into one text summary for profiles/.
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived counters); per MI355X_MICROARCH.md (HBM) FETCH_SIZE on
gfx950 reports half of the bytes of a wide coalesced streaming read, so the corrected read traffic is 2 x."""
import collections
import csv
import glob
import json
import re
import sys

FAMILY = {"mcmc": ["mcmc_kernel"], "phmm": ["phmm_kernel", "finalize_kernel"],
          "polish": ["sum_tables_kernel", "select_edits_kernel", "rethread_kernel", "commit_kernel", "band_prep_kernel"],
          "filter": ["homop_kernel", "chunk_tables_kernel", "column_filter_kernel", "pick_kernel"]}


def short(name):
    m = re.search(r"::(\w+)\(", name)
    return m.group(1) if m else name.split("(")[0]


def main(root, json_path=None):
    out = []
    pmc = {}
    stats = glob.glob(f"{root}/prof_stats/*/*_kernel_stats.csv")
    out.append("== rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 16 --warmup 1 --no-cpu-baseline ==")
    out.append("(20 hot-path passes over 500-chunk batches, 4 batches in flight: 4 warm-up + 16 timed)")
    out.append(f"{'kernel':28s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>12s} {'pct':>7s}")
    for f in stats:
        for r in csv.DictReader(open(f)):
            out.append(f"{short(r['Name']):28s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:12.3f} "
                       f"{float(r['AverageNs'])/1e6:12.3f} {float(r['Percentage']):7.3f}")
    for label, pat in (("FETCH_SIZE", "prof_fetch"), ("WRITE_SIZE", "prof_write")):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for f in glob.glob(f"{root}/{pat}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
        out.append("")
        out.append(f"== rocprofv3 --kernel-trace --pmc {label} (own pass), KiB ==")
        out.append(f"{'kernel':28s} {'launches':>8s} {'sum_KiB':>16s} {'per_launch_KiB':>16s}")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            out.append(f"{k:28s} {n:8d} {v:16.1f} {v / n:16.1f}")
            pmc.setdefault(k, {})[f"{label}_KiB_per_launch"] = v / n
            pmc[k]["launches_in_profile"] = n
    print("\n".join(out))
    if json_path:  # what bench.py reads back as roofline.traffic
        json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py "
                              "--steps 16 --warmup 1 --no-cpu-baseline",
                   "workload": "cfg2_ont_diploid_500x60x2kbp",
                   "note": "KiB per launch; FETCH_SIZE on gfx950 reports half of the bytes of wide coalesced reads "
                           "(MI355X_MICROARCH.md HBM): hbm_bytes = (2*FETCH + WRITE)*1024",
                   "kernel_family": FAMILY, "kernels": pmc}, open(json_path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else None)
