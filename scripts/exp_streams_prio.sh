#!/bin/bash
# Diagnostic (GPU box): the bench line under other slice counts (--streams) and with raised wave priority in the chain
# workgroups (experiment builds -DJTK_MCMC_PRIO_CONSUMER / _PRODUCER), headline and cfg 2.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
COMMON="--no-cpu-baseline --no-e2e --no-shard8"
run() {  # name lib workload streams steps
  local name=$1 lib=$2 wl=$3 st=$4 steps=$5
  if [ -n "$lib" ]; then export JTK_LC_LIB=$PWD/jtk_amd/_build/exp_$lib/libjtk_lc_$lib.so; else unset JTK_LC_LIB; fi
  timeout 300 python3 bench.py --workload $wl --streams $st --steps $steps --warmup 2 $COMMON > $O/xs_$name.json 2> $O/xs_$name.err
  python3 - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/xs_%s.json" % name).read().strip().splitlines()[-1])
    print("%-18s value %8.1f ms/step %8.1f serial %s" % (name, d["value"], d["ms_per_step"],
          json.dumps({k: round(v) for k, v in d["roofline"]["serial_pass"]["kernel_ms"].items()})), flush=True)
except Exception as e:
    print(name, "no line:", e, flush=True)
PY
}
H=cfg3_ont_diploid_2500x60x2kbp
C2=cfg2_ont_diploid_500x60x2kbp
run h_s6 "" $H 6 8
run h_s8 "" $H 8 8
run h_s10 "" $H 10 8
run h_c3 prio_c3 $H 6 8
run h_c3p3 prio_c3p3 $H 6 8
run c2_s6 "" $C2 6 8
run c2_s4 "" $C2 4 8
run c2_s8 "" $C2 8 8
run c2_c3 prio_c3 $C2 6 8
run c2_c3p3 prio_c3p3 $C2 6 8
unset JTK_LC_LIB
