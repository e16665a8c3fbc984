#!/bin/bash
# GPU box: what a rank of an N-GPU strong-scaling run sees (2500 / N chunks) with different slice counts and numbers of
# hardware queues (HIP maps streams onto GPU_MAX_HW_QUEUES queues, 4 by default: streams that share one run in order)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ${@:-"312 4 4" "312 8 8" "312 8 16" "312 16 16" "2500 4 8" "2500 6 8" "2500 8 8" "2500 8 16"}; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 python bench.py --chunks $1 --streams $2 --no-cpu-baseline --no-e2e --steps 8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('chunks $1 streams $2 hwq $3:', round(d['value'], 1), 'chunks/s', round(d['ms_per_step'], 1), 'ms/step')
"
done
