#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace --output-format csv directory and prints what a summary of average durations cannot show:
who is on the device WHEN.  Per kernel name: launches, summed duration, the union of its busy intervals; per queue (= one
slice's stream): the gaps between the end of one dispatch and the start of the next, by the kernel that follows (time a slice
spends waiting for the host, for an event or for wave slots); and in 20 ms bins, how many pair-HMM, polish and chain kernels are
running.

    python3 scripts/trace_timeline.py gpurun_out/trace_r6
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    # (the kernels live in anonymous namespaces: "(anonymous namespace)::mcmc_kernel_light(ChunkMeta const*, ...")
    for key in ("mcmc_kernel_light", "mcmc_kernel_huge", "mcmc_kernel", "phmm_pair_kernel", "phmm_wide_kernel", "phmm_kernel",
                "sum_final_kernel", "finalize_kernel", "rethread_kernel", "band_prep_kernel", "commit_kernel", "select_edits_kernel",
                "column_filter_kernel", "pick_kernel", "chunk_tables_kernel", "homop_kernel", "chain_split_kernel", "reset_pass_kernel",
                "gather_kernel", "out_len_kernel", "encode_reads_kernel", "copyBuffer", "fillBuffer"):
        if key in name:
            return key
    return name.split("(")[0][-40:] or name[:40]


def union_ms(iv):
    iv = sorted(iv)
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot / 1e6


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel_trace.csv under", d)
        return 1
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), short(r["Kernel_Name"]),
                             int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    wall = (t1 - t0) / 1e6
    print("dispatches %d, wall %.1f ms, queues %d" % (len(rows), wall, len({r[2] for r in rows})))
    by = collections.defaultdict(list)
    for a, b, q, k, g, w in rows:
        by[k].append((a, b))
    print("%-22s %6s %10s %10s %8s" % ("kernel", "calls", "sum_ms", "union_ms", "avg_ms"))
    for k, iv in sorted(by.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
        s = sum(b - a for a, b in iv) / 1e6
        print("%-22s %6d %10.1f %10.1f %8.2f" % (k, len(iv), s, union_ms(iv), s / len(iv)))
    fam = {"phmm": ("phmm_kernel", "phmm_pair_kernel", "phmm_wide_kernel"), "sum/final": ("sum_final_kernel", "finalize_kernel"),
           "polish": ("rethread_kernel", "band_prep_kernel", "commit_kernel", "select_edits_kernel", "reset_pass_kernel"),
           "chain": ("mcmc_kernel_light", "mcmc_kernel", "mcmc_kernel_huge")}
    for name, ks in fam.items():
        iv = [x for k in ks for x in by.get(k, [])]
        print("union of %-10s %8.1f ms of %8.1f (%.0f %%)" % (name, union_ms(iv), wall, 100 * union_ms(iv) / wall))
    print("union of everything %8.1f ms" % union_ms([(a, b) for a, b, *_ in rows]))
    # gaps inside a queue, by the kernel that FOLLOWS the gap
    gaps = collections.defaultdict(lambda: [0, 0.0, 0.0])
    perq = collections.defaultdict(list)
    for r in rows:
        perq[r[2]].append(r)
    for q, rs in perq.items():
        rs.sort()
        for prev, nxt in zip(rs, rs[1:]):
            g = (nxt[0] - prev[1]) / 1e6
            if g > 0:
                e = gaps[(prev[3], nxt[3])]
                e[0] += 1
                e[1] += g
                e[2] = max(e[2], g)
    print("gaps inside a queue (end of one dispatch -> start of the next), the twelve largest sums:")
    for (p, n), (c, s, m) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  %-22s -> %-22s n %5d sum %9.1f ms  mean %7.3f  max %7.2f" % (p, n, c, s, s / c, m))
    # who is running, in 20 ms bins (mean number of dispatches of the family in flight)
    BIN = 20e6
    nb = int((t1 - t0) / BIN) + 1
    run = {name: [0.0] * nb for name in fam}
    for a, b, q, k, g, w in rows:
        for name, ks in fam.items():
            if k in ks:
                i0, i1 = int((a - t0) / BIN), int((b - t0) / BIN)
                for i in range(i0, i1 + 1):
                    lo, hi = max(a, t0 + i * BIN), min(b, t0 + (i + 1) * BIN)
                    if hi > lo:
                        run[name][i] += (hi - lo) / BIN
    print("mean dispatches in flight per 20 ms bin (phmm | sum/final | polish | chain), every 2nd bin:")
    line = []
    for i in range(0, nb, 2):
        line.append("%.1f|%.1f|%.1f|%.1f" % tuple(run[n][i] for n in ("phmm", "sum/final", "polish", "chain")))
    for i in range(0, len(line), 10):
        print("  " + "  ".join(line[i:i + 10]))
    # pair-HMM launches: duration against the work they carried is not in the trace; print the distribution of durations
    ph = sorted((b - a) / 1e6 for a, b in by.get("phmm_kernel", []))
    if ph:
        print("phmm_kernel durations ms: min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f" % (ph[0], ph[len(ph) // 4], ph[len(ph) // 2], ph[3 * len(ph) // 4], ph[-1]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
