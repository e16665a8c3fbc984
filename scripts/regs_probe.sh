#!/bin/bash
# GPU box: does a pair-HMM wave fit beside a chain wave?  libjtk_lc_base.so = phmm_kernel at 160 registers, libjtk_lc.so = 152
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
show() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['value'], 1), 'chunks/s', round(d['ms_per_step'], 1), 'ms/step', {k: round(v) for k, v in d['roofline']['serial_pass']['kernel_ms'].items()})
"; }
for lib in "$@"; do
  JTK_LC_LIB=$PWD/jtk_amd/_build/$lib python bench.py --no-cpu-baseline --no-e2e --steps 6 2>/dev/null | show $lib
done
