#!/bin/bash
# Diagnostic (GPU box): the chain kernel's time on one 625-chunk (and 125-chunk) slice of the headline workload for
# prebuilt variants of the library: chain_time_variants.sh name:@lib.so ...
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/chain_time.py <<'PY'
import sys, time
import torch  # noqa
sys.path.insert(0, ".")
from jtk_amd import api, batch as jb, synth
for n in (625, 125):
    b, cfg = synth.make_batch("ont_diploid", n)
    p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
    with api.Session(p, b) as s:
        for it in range(2):
            s.run()
            t = api.last_timing()
        print("CHAINMS n=%d" % n, {k: round(v, 1) for k, v in t["kernel_ms"].items()})
PY
for spec in "$@"; do
  name=${spec%%:*}; lib=${spec#*:@}
  if [ "$lib" != "$spec" ] && [ -n "$lib" ]; then export JTK_LC_LIB=$lib; else unset JTK_LC_LIB; fi
  echo "== $name"; python3 /tmp/chain_time.py 2>&1 | grep CHAINMS
done
