#!/bin/bash
# Diagnostic (GPU box): the default bench line for builds of the library with extra compile flags, beside the product.
# usage: bench_variants.sh "name1:-DFLAG ..." "name2:@prebuilt.so" "name3:"   (an empty flag list = the product library itself)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  if [ "${flags:0:1}" = "@" ]; then
    export JTK_LC_LIB=${flags:1}
  elif [ -n "$flags" ]; then
    export JTK_LC_LIB=$(python3 -c "import jtk_amd.build as b; print(b.build_experiment('$name', '''$flags'''))" 2>gpurun_out/bv_$name.build.err) || { echo "$name: build failed"; tail -3 gpurun_out/bv_$name.build.err; continue; }
  else
    unset JTK_LC_LIB
  fi
  python3 bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-e2e ${EXTRA:-} > gpurun_out/bv_$name.json 2> gpurun_out/bv_$name.err
  python3 - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/bv_%s.json" % name).read().strip().splitlines()[-1])
    sp = d.get("roofline", {}).get("secondary", {})
    print("%-12s value %.1f ms/step %.1f | stage_e2e warm %s | shard8 ms %s | serial kernel ms %s" % (
        name, d["value"], d["ms_per_step"],
        json.dumps((d.get("stage_e2e") or {}).get("chunks_per_s_warm")),
        json.dumps((d.get("shard8_projection") or {}).get("ms_per_step")),
        json.dumps({k: round(v) for k, v in d["roofline"]["serial_pass"]["kernel_ms"].items()})))
except Exception as e:
    print(name, "no line:", e)
PY
done
unset JTK_LC_LIB
