#!/bin/bash
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
JTK_EXTRA_HIPCC_FLAGS="-DJTK_MCMC_STATS" python3 -c "import jtk_amd.build as b; b.build(force=True)" 2>&1 | grep -i ' error'
python3 /tmp/tabrun.py 2 > gpurun_out/tabstat_raw.txt 2>&1
grep MCMCMS gpurun_out/tabstat_raw.txt
grep -c TABSTAT gpurun_out/tabstat_raw.txt
python3 - <<'PY'
import re
tot = 0
for line in open("gpurun_out/tabstat_raw.txt"):
    m = re.search(r"cyc_total (\d+)", line)
    if m: tot += int(m.group(1))
print("sum of chain cycles over both chunks", tot, "-> per chunk s at 2.4GHz", tot/2/2.4e9)
PY
grep -E "K2PROD|K2WAIT" gpurun_out/tabstat_raw.txt | head -5
python3 -c "import jtk_amd.build as b; b.build(force=True)" 2>&1 | grep -i ' error'
