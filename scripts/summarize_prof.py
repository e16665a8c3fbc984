#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by scripts/profile_bench.sh into one text summary (stdout) and the JSON bench.py
reads back as roofline.traffic (gpurun_out/prof_traffic_<tag>.json; copy both into profiles/).
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived counters); per MI355X_MICROARCH.md (HBM) FETCH_SIZE on gfx950
reports half of the bytes of a wide coalesced streaming read, so the corrected read traffic is 2 x."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jtk_amd import build as jbuild  # noqa: E402

FAMILY = {"mcmc": ["mcmc_kernel_light", "mcmc_kernel", "chain_split_kernel"], "phmm": ["phmm_kernel", "phmm_pair_kernel", "phmm_wide_kernel", "finalize_kernel", "sum_final_kernel"],
          "polish": ["select_edits_kernel", "rethread_kernel", "commit_kernel", "band_prep_kernel"],
          "filter": ["homop_kernel", "chunk_tables_kernel", "column_filter_kernel", "column_filter_fused_kernel", "pick_kernel"]}


def short(name):
    m = re.search(r"::(\w+)\(", name)
    return m.group(1) if m else name.split("(")[0]


def main(root, tag):
    cmd = open(f"{root}/prof_cmd_{tag}.txt").read().strip()
    sha = open(f"{root}/prof_libsha_{tag}.txt").read().strip()
    workload = "cfg3_ont_diploid_2500x60x2kbp"
    m = re.search(r"--workload (\S+)", cmd)
    if m:
        workload = m.group(1)
    # passes of the workload in the profiled command: warm-up + timed steps + the one serial pass after them
    ms, mw = re.search(r"--steps (\d+)", cmd), re.search(r"--warmup (\d+)", cmd)
    n_passes = (int(ms.group(1)) if ms else 6) + (int(mw.group(1)) if mw else 1) + 1
    clean = "--no-shard8" in cmd  # otherwise the launches of the 313-chunk shard runs are in the sums too
    out = [f"== rocprofv3 --kernel-trace --stats -- {cmd} ==", f"(library sha256[:16] {sha}; kernel sources {jbuild.source_sha16()}; workload {workload})",
           f"{'kernel':28s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>12s} {'pct':>7s}"]
    pmc = {}
    for f in glob.glob(f"{root}/prof_stats_{tag}/**/*_kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out.append(f"{short(r['Name']):28s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:12.3f} "
                       f"{float(r['AverageNs'])/1e6:12.3f} {float(r['Percentage']):7.3f}")
            pmc.setdefault(short(r["Name"]), {})["avg_ms"] = float(r["AverageNs"]) / 1e6
    for label, pat in (("FETCH_SIZE", f"prof_fetch_{tag}"), ("WRITE_SIZE", f"prof_write_{tag}")):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for f in glob.glob(f"{root}/{pat}/**/*_counter_collection.csv", recursive=True):
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
            pmc[k][f"{label}_KiB_per_pass"] = v / n_passes
            pmc[k]["launches_in_profile"] = n
    out.append("")
    out.append(f"== HBM bytes per pass of the workload ({n_passes} passes in the profile; (2 x FETCH_SIZE + WRITE_SIZE) x 1024"
               + ("" if clean else "; NOT clean: the command ran the shard8 leg too") + ") ==")
    fam_pass = {}
    for fam, kernels in FAMILY.items():
        b = sum((2.0 * pmc.get(k, {}).get("FETCH_SIZE_KiB_per_pass", 0.0) + pmc.get(k, {}).get("WRITE_SIZE_KiB_per_pass", 0.0)) * 1024.0
                for k in kernels)
        fam_pass[fam] = b
        out.append(f"{fam:10s} {b / 1e9:12.2f} GB per pass")
    print("\n".join(out))
    json.dump({"command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- {cmd}",
               "workload": workload, "lib_sha16": sha, "src_sha16": jbuild.source_sha16(),
               "note": "KiB per launch; FETCH_SIZE on gfx950 reports half of the bytes of wide coalesced reads "
                       "(MI355X_MICROARCH.md HBM): hbm_bytes = (2*FETCH + WRITE)*1024",
               "passes_in_profile": n_passes, "full_workload_launches_only": clean,
               "family_bytes_per_pass": fam_pass,
               "kernel_family": FAMILY, "kernels": pmc}, open(f"{root}/prof_traffic_{tag}.json", "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "r04")
