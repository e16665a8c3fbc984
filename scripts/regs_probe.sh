#!/bin/bash
# GPU box: the default bench on experiment builds of the library (jtk_amd/_build/<name>.so [ENV=VALUE ...])
# usage: regs_probe.sh "libA.so" "libB.so JTK_MCMC_JUMP_GLOBAL=1" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
show() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['value'], 1), 'chunks/s', round(d['ms_per_step'], 1), 'ms/step', {k: round(v) for k, v in d['roofline']['serial_pass']['kernel_ms'].items()}, 'steps_agree', d.get('steps_agree'))
"; }
for spec in "$@"; do
  set -- $spec
  lib=$1; shift
  env JTK_LC_LIB=$PWD/jtk_amd/_build/$lib "$@" python bench.py --no-cpu-baseline --no-e2e --steps 6 2>/dev/null | show "$lib $*"
done
