#!/bin/bash
# GPU box: the default bench with the chain kernel's jump table in LDS (default) and in global memory
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
show() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['value'], 1), 'chunks/s', round(d['ms_per_step'], 1), 'ms/step', {k: round(v) for k, v in d['roofline']['serial_pass']['kernel_ms'].items()})
"; }
JTK_MCMC_JUMP_LDS=1 python bench.py --no-cpu-baseline --no-e2e --steps 6 2>/dev/null | show lds
python bench.py --no-cpu-baseline --no-e2e --steps 6 2>/dev/null | show global
python bench.py --no-cpu-baseline --no-e2e --steps 6 --streams 6 2>/dev/null | show global6
