#!/bin/bash
# Diagnostic (GPU box), round 6: the kernel trace (profiles/r06_trace_summary.txt) shows the six slices of the headline step in
# LOCKSTEP -- all in their pair-HMM rounds together, then all in their chain kernels together, 200-300 ms per step in which the
# device runs nothing but chain workgroups.  Start offsets between the slices, driver's step count.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/stagger_r6.txt
echo "== $(date -u +%FT%TZ)" >> $OUT
short() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.1f ms_per_step %.1f' % (d['value'], d['ms_per_step']))
"; }
for cfg in "6 0" "6 60" "6 100" "6 150" "8 80" "4 150" "12 50"; do
  set -- $cfg
  echo "-- streams $1 stagger $2 ms: $(timeout 600 python3 bench.py --steps 20 --warmup 3 --streams $1 --stagger-ms $2 --no-cpu-baseline --no-shard8 --no-e2e 2>/dev/null | short)" | tee -a $OUT
done
