#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc CSVs written by scripts/pmc_phmm.sh: per kernel, the mean per launch of every counter
collected, the kernel's mean duration (the --stats pass) and the ratios the issue-limit argument rests on.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); SQ_BUSY_CYCLES
is per shader engine; SQ_INSTS_* count wave-instructions."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"::(\w+)\(", name)
    return m.group(1) if m else name.split("(")[0]


def main(root, tag):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(f"{root}/pmc_{tag}_g*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            e = per[short(r["Kernel_Name"])][r["Counter_Name"]]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    dur = {}
    for f in glob.glob(f"{root}/pmc_{tag}_stats/**/*_kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
    cmd = open(f"{root}/pmc_cmd_{tag}.txt").read().strip()
    sha = open(f"{root}/pmc_libsha_{tag}.txt").read().strip()
    print(f"== rocprofv3 --kernel-trace --pmc <group> (one pass per group) -- {cmd} ==")
    print(f"(library sha256[:16] {sha})")
    out = {"command": cmd, "lib_sha16": sha, "kernels": {}}
    for k in sorted(per, key=lambda k: -per[k].get("SQ_WAVE_CYCLES", [0, 0])[1]):
        c = {name: v / n for name, (n, v) in per[k].items()}
        if k in dur:
            c["avg_ms"] = dur[k][1]
        out["kernels"][k] = c
        print(f"\n-- {k}" + (f" (avg {dur[k][1]:.3f} ms over {dur[k][0]} launches)" if k in dur else ""))
        for name in sorted(c):
            print(f"   {name:28s} {c[name]:18.1f}")
        g = c.get
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            for a in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                      "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
                if g(a) is not None:
                    print(f"   {a + ' / SQ_WAVE_CYCLES':44s} {g(a) / wc:8.3f}")
        if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
            tot = sum(g(x, 0.0) for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                                          "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"))
            print(f"   {'wave-instructions per launch (sum of classes)':44s} {tot:14.0f}")
            for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
                      "SQ_INSTS_BRANCH"):
                if g(x) is not None:
                    print(f"   {'  share ' + x:44s} {g(x) / tot:8.3f}")
        if g("SQ_LDS_IDX_ACTIVE"):
            print(f"   {'SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE':44s} {g('SQ_LDS_BANK_CONFLICT', 0.0) / g('SQ_LDS_IDX_ACTIVE'):8.3f}")
        if g("SQC_ICACHE_REQ"):
            print(f"   {'SQC_ICACHE_MISSES / SQC_ICACHE_REQ':44s} {g('SQC_ICACHE_MISSES', 0.0) / g('SQC_ICACHE_REQ'):8.4f}")
    json.dump(out, open(f"{root}/pmc_issue_{tag}.json", "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "r03")
