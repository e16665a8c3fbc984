#!/bin/bash
# Diagnostic (GPU box): per-chunk event statistics of the table-driven chain on 4-copy pile-ups; lists the slowest chunks.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-48}
cat > /tmp/tabrun.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from jtk_amd import api, batch as jb, synth
n = int(sys.argv[1])
b, cfg = synth.make_batch("ont_4copy", n)
p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
with api.Session(p, b) as s:
    s.run()
    t = api.last_timing()
print("MCMCMS", t["kernel_ms"]["mcmc"])
PY
export JTK_LC_LIB=$(python3 -c "import jtk_amd.build as b; print(b.build_experiment('probe1', '''-DJTK_MCMC_STATS'''))") || exit 1
python3 /tmp/tabrun.py $N > gpurun_out/tabstat_raw.txt 2>&1
grep MCMCMS gpurun_out/tabstat_raw.txt
python3 - <<'PY'
import re, collections
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0]*9))
for line in open("gpurun_out/tabstat_raw.txt"):
    m = re.search(r"TABSTAT chunk (\d+) K (\d+) n (\d+) D (\d+) steps (\d+) fast (\d+) events (\d+) accepts (\d+) reloads (\d+) scalars (\d+) cyc_rebuild (\d+) cyc_event (\d+) cyc_total (\d+)", line)
    if m:
        v = [int(x) for x in m.groups()]
        a = per[v[0]][(v[1], v[3])]
        for i in range(9): a[i] += v[4 + i]
tot = {c: sum(a[8] for a in ks.values()) for c, ks in per.items()}
order = sorted(tot, key=lambda c: -tot[c])
import statistics
print("chunks", len(order), "mean s", statistics.mean(tot.values())/2.4e9, "max s", max(tot.values())/2.4e9)
for c in order[:6] + order[-2:]:
    print("chunk", c, "total %.2f s" % (tot[c]/2.4e9))
    for (K, D), a in sorted(per[c].items()):
        st = a[0]
        print("   K %d D %d: %.0f cyc/step; events %.2f%% accepts %.2f%% reloads %.1f%%; event %.0f cyc (rebuild %.0f)" % (
            K, D, a[8]/st, 100*a[2]/st, 100*a[3]/st, 100*a[4]/st, a[7]/max(1,a[2]), a[6]/max(1,a[3])))
PY
unset JTK_LC_LIB   # the product library was never touched
