#!/bin/bash
# Diagnostic (GPU box): the bench line's headline, stage_e2e and e2e blocks for the product library and for prebuilt variants:
# stage_ab.sh name:@lib.so ...   ("product:" = the product library)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
  name=${spec%%:*}; lib=${spec#*:@}
  if [ "$lib" != "$spec" ] && [ -n "$lib" ]; then export JTK_LC_LIB=$lib; else unset JTK_LC_LIB; fi
  python3 bench.py --steps ${STEPS:-4} --warmup 2 --no-cpu-baseline --no-shard8 > gpurun_out/sab_$name.json 2> gpurun_out/sab_$name.err
  python3 - "$name" <<'PY'
import json, sys
name = sys.argv[1]
d = json.loads(open("gpurun_out/sab_%s.json" % name).read().strip().splitlines()[-1])
se = d["stage_e2e"]; w = se["warm"]
print("%-10s value %.1f | stage_e2e warm %.1f chunks/s: cluster_chunks %.0f ms, chain summed %.0f ms | e2e %.2f s | serial chain %.0f ms" % (
    name, d["value"], se["chunks_per_s_warm"], w["cluster_chunks_ms"], w["cluster_chunks_detail"]["kernel_ms_summed_over_slices"]["mcmc"],
    d["e2e"]["seconds"], d["roofline"]["serial_pass"]["kernel_ms"]["mcmc"]))
PY
done
unset JTK_LC_LIB
