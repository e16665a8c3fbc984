#!/bin/bash
# Diagnostic (GPU box): where the table-driven chain (mcmc_chain_tab) spends its cycles, on full-size 4-copy pile-ups,
# (the one-proposal-per-iteration chain it used to be timed against left the product sources in round 4: git history).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/tabrun.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from jtk_amd import api, batch as jb, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
b, cfg = synth.make_batch("ont_4copy", n)
p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
with api.Session(p, b) as s:
    s.run()
    t = api.last_timing()
    r = s.fetch_results()
print("MCMCMS", t["kernel_ms"]["mcmc"], "k", r["result"]["cluster_num"].tolist(), "D", r["result"]["n_variants"].tolist())
PY
python3 /tmp/tabrun.py 8 2>&1 | grep MCMCMS
export JTK_LC_LIB=$(python3 -c "import jtk_amd.build as b; print(b.build_experiment('probe2', '''-DJTK_MCMC_STATS'''))") || exit 1
python3 /tmp/tabrun.py 2 > gpurun_out/tabstat_raw.txt 2>&1
python3 - <<'PY'
import re, collections
acc = collections.defaultdict(lambda: [0] * 10)
for line in open("gpurun_out/tabstat_raw.txt"):
    m = re.search(r"TABSTAT K (\d+) n (\d+) D (\d+) steps (\d+) fast (\d+) events (\d+) accepts (\d+) reloads (\d+) scalars (\d+) cyc_rebuild (\d+) cyc_event (\d+) cyc_total (\d+)", line)
    if m:
        v = [int(x) for x in m.groups()]
        a = acc[(v[0], v[1], v[2])]
        a[0] += 1
        for i in range(9):
            a[i + 1] += v[3 + i]
for key, a in sorted(acc.items()):
    st = a[1]
    print("K %d n %d D %d: %d chains; per step: %.1f cycles total; fast %.1f%% events %.2f%% accepts %.2f%% reloads %.2f%% scalars %.3f%%; "
          "rebuild %.0f cyc each, event (incl. rebuild) %.0f cyc each"
          % (*key, a[0], a[9] / st, 100.0 * a[2] / st, 100.0 * a[3] / st, 100.0 * a[4] / st, 100.0 * a[5] / st, 100.0 * a[6] / st,
             a[7] / max(1, a[4]), a[8] / max(1, a[3])))
PY
unset JTK_LC_LIB   # the product library was never touched
