#!/bin/bash
# Diagnostic (GPU box), round 6: the pair-HMM gate (session.hip: PhaseGate) -- at most n batches per device in their pair-HMM
# rounds at a time -- against the free-for-all of rounds 1-5 (JTK_LC_PHASE_SLOTS=0), driver's step count.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/gate_r6.txt
echo "== $(date -u +%FT%TZ)" >> $OUT
short() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d.get('e2e') or {}
print('value %.1f ms_per_step %.1f e2e %s' % (d['value'], d['ms_per_step'], round(e.get('seconds',0),3)))
"; }
for cfg in "6 0" "6 2" "6 3" "6 4" "8 3" "12 3" "12 4" "6 1"; do
  set -- $cfg
  echo "-- streams $1 gate slots $2: $(JTK_LC_PHASE_SLOTS=$2 timeout 600 python3 bench.py --steps 12 --warmup 2 --streams $1 --no-cpu-baseline --no-shard8 --no-e2e 2>/dev/null | short)" | tee -a $OUT
done
for g in 0 2 3; do
echo "-- one-shot call, gate slots $g: $(JTK_LC_PHASE_SLOTS=$g timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-shard8 2>/dev/null | short)" | tee -a $OUT
done
for sl in 6 8; do
echo "-- one-shot call, $sl slices, gate slots 3: $(JTK_LC_SLICES=$sl JTK_LC_PHASE_SLOTS=3 timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-shard8 2>/dev/null | short)" | tee -a $OUT
done
