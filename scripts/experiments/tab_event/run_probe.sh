#!/bin/bash
# GPU box: the K-way table chain's event, piece by piece, on two full-size 4-copy pile-ups (library built by build_probe.py).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/tabx.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from jtk_amd import api, batch as jb, synth
b, cfg = synth.make_batch("ont_4copy", 2)
p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
with api.Session(p, b) as s:
    s.run()
PY
JTK_LC_LIB=$PWD/jtk_amd/_build/exp_tabx/libjtk_lc_tabx.so timeout 400 python3 /tmp/tabx.py > gpurun_out/tabx_raw.txt 2>&1
python3 - <<'PY'
import re, collections
acc = collections.defaultdict(lambda: [0] * 17)
pat = re.compile(r"TABX K (\d+) n (\d+) D (\d+) steps (\d+) events (\d+) accepts (\d+) setup (\d+) getlk (\d+) bern (\d+) book (\d+) publish (\d+) hops (\d+) exact_exp (\d+) event (\d+) total (\d+) acc_state (\d+) acc_label (\d+) new_max (\d+)")
for line in open("gpurun_out/tabx_raw.txt"):
    m = pat.search(line)
    if m:
        v = [int(x) for x in m.groups()]
        a = acc[(v[0], v[1], v[2])]
        a[0] += 1
        for i in range(15):
            a[i + 1] += v[3 + i]
print("mcmc_chain_tab, cycles per EVENT (an event = one proposal redone exactly); n reads, D columns")
for key, a in sorted(acc.items()):
    ev = max(1, a[2])
    print("K %d n %d D %d: %d chains, %.1f %% of the proposals are events, %.1f %% accepted | per event: set-up %.0f, get_lk %.0f, Bernoulli %.0f "
          "(exact exp in %.1f %% of the events), accept / flip-back %.0f [state + size terms %.0f, label store + sync %.0f, best-state copy in %.1f %% of the accepts], republish %.0f, hop words / window %.0f | event total %.0f | chain total %.0f cycles per proposal"
          % (*key, a[0], 100.0 * a[2] / a[1], 100.0 * a[3] / a[1], a[4] / ev, a[5] / ev, a[6] / ev, 100.0 * a[10] / ev, a[7] / ev, a[13] / max(1, a[3]), a[14] / max(1, a[3]), 100.0 * a[15] / max(1, a[3]), a[8] / ev, a[9] / ev,
             a[11] / ev, a[12] / a[1]))
PY
