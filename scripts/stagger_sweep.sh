#!/bin/bash
# GPU box: the default bench workload with different slice counts / start offsets (one line each)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for cfg in ${@:-"4 0" "3 0" "5 0" "6 0" "8 0" "10 0"}; do
  set -- $cfg
  python bench.py --streams $1 --stagger-ms $2 --no-cpu-baseline --no-e2e --steps 6 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('streams $1 stagger $2:', round(d['value'], 1), 'chunks/s', round(d['ms_per_step'], 1), 'ms/step')
"
done
