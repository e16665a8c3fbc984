#!/bin/bash
# Diagnostic (GPU box): where the generic chain (K > 2, or n > 127) spends its cycles, on full-size 4-copy pile-ups.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
export JTK_LC_LIB=$(python3 -c "import jtk_amd.build as b; print(b.build_experiment('probe1', '''-DJTK_MCMC_STATS'''))") || exit 1
python3 - > gpurun_out/genstat_raw.txt 2>&1 <<'PY'
import torch
from jtk_amd import api, batch as jb, synth
b, cfg = synth.make_batch("ont_4copy", 4)
p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
api.cluster_chunks(p, b)
PY
python3 - <<'PY'
import re, collections
acc = collections.defaultdict(lambda: [0] * 8)
for line in open("gpurun_out/genstat_raw.txt"):
    for m in re.finditer(r"GENSTAT K (\d+) n (\d+) D (\d+) steps (\d+) draws (\d+) flip (\d+) approx (\d+) decide (\d+) tail (\d+) exact (\d+)", line):
        v = [int(x) for x in m.groups()]
        a = acc[(v[0], v[1], v[2])]
        a[0] += 1
        for i in range(7):
            a[i + 1] += v[3 + i]
for key, a in sorted(acc.items()):
    st = a[1]
    print("K %d n %d D %d: %d chains, per step: draws %.0f flip %.0f approx %.0f decide %.0f tail %.0f cycles; exact evaluations %.2f%%"
          % (*key, a[0], a[2] / st, a[3] / st, a[4] / st, a[5] / st, a[6] / st, 100.0 * a[7] / st))
PY
unset JTK_LC_LIB   # the product library was never touched
