#!/bin/bash
# Diagnostic (GPU box): the chain kernels on full-size 4-copy pile-ups (cfg 4: 160 reads): device time of the product library on 8
# pile-ups, then -- with the -DJTK_MCMC_STATS library given as $1 -- per candidate K what mcmc_chain_tab's steps are made of,
# including how many certainly rejected steps leave a rounding residue in a sum (those cannot be skipped).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 scripts/experiments/tab_event/chain_ms.py 8 2>&1 | grep MCMCMS
[ -n "${1:-}" ] || exit 0
JTK_LC_LIB=$1 python3 scripts/experiments/tab_event/chain_ms.py 2 > gpurun_out/tabstat3_raw.txt 2>&1
python3 - <<'PY'
import re, collections
acc = collections.defaultdict(lambda: [0] * 16)
pat = re.compile(r"TABSTAT chunk \d+ K (\d+) n (\d+) D (\d+) steps (\d+) fast (\d+) events (\d+) accepts (\d+) reloads (\d+) scalars (\d+) cyc_rebuild (\d+) cyc_event (\d+) cyc_total (\d+) residues (\d+) cyc_wload (\d+) cyc_hopw (\d+) uncert (\d+) cyc_fast (\d+) fast_entries (\d+)")
for line in open("gpurun_out/tabstat3_raw.txt"):
    for m in pat.finditer(line):
        v = [int(x) for x in m.groups()]
        a = acc[(v[0], v[1], v[2])]
        a[0] += 1
        for i in range(15):
            a[i + 1] += v[3 + i]
for key, a in sorted(acc.items()):
    st = a[1]
    print("mcmc_chain_tab K %d n %d D %d: %d chains; per step %.1f cycles; fast %.1f %% (of which leave a residue: %.3f %% of all steps) events %.2f %% accepts %.2f %% "
          "window reloads %.2f %%; republish %.0f cycles each, event (incl. republish) %.0f cycles each; a window move: load %.0f + hop words %.0f cycles, "
          "%.2f uncertified columns; rejected steps: %.0f cycles each in blocks of %.1f"
          % (*key, a[0], a[9] / st, 100.0 * a[2] / st, 100.0 * a[10] / st, 100.0 * a[3] / st, 100.0 * a[4] / st, 100.0 * a[5] / st,
             a[7] / max(1, a[4]), a[8] / max(1, a[3]), a[11] / max(1, a[5]), a[12] / max(1, a[5]), a[13] / max(1, a[5]), a[14] / max(1, a[2]), a[2] / max(1, a[15])))
k2 = [l for l in open("gpurun_out/tabstat3_raw.txt") if "K2STAT" in l]
print("diploid-chain workgroups (K2STAT lines):", len(k2))
PY
