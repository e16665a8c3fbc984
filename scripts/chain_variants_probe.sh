#!/bin/bash
# Diagnostic (GPU box): (a) the light chain kernel taking pile-ups up to 127 reads (product) against up to 63 (experiment build) on
# Poisson(60) coverage and on the headline; (b) the general chain kernel on its own stream (product) against the same stream
# as the light one (JTK_MCMC_SIDE=0); (c) the device's shared stripe set against one set per session (JTK_STRIPE_SHARED=0).
# Same box, alternating.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
L63=$(python3 -c "import jtk_amd.build as b; print(b.build_experiment('light63', '-DJTK_LIGHT_MAX_READS=63u'))") || exit 1
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k: round(v) for k, v in d['roofline']['serial_pass']['kernel_ms'].items()})"; }
for rep in 1 2; do
  echo "== poisson, light <= 127 reads"; timeout 300 python3 scripts/poisson_coverage_bench.py 500 | tail -1
  echo "== poisson, light <= 63 reads";  JTK_LC_LIB=$L63 timeout 300 python3 scripts/poisson_coverage_bench.py 500 | tail -1
  timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "headline side-stream"
  JTK_MCMC_SIDE=0 timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "headline one-stream "
  JTK_LC_LIB=$L63 timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "headline light63   "
  JTK_STRIPE_SHARED=0 timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "headline own stripes"
done
