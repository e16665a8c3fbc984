#!/bin/bash
# Diagnostic (GPU box): the chain launch as (a) the general kernel alone (JTK_MCMC_SPLIT=0), (b) light at 168 registers (three
# waves per SIMD: product), (c) light at 176 registers (two per SIMD: -DJTK_LIGHT_176) on cfg 3, cfg 5 and cfg 2.  Same box.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
L176=$(python3 -c "import jtk_amd.build as b; print(b.build_experiment('light176', '-DJTK_LIGHT_176'))") || exit 1
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k: round(v) for k, v in d['roofline']['serial_pass']['kernel_ms'].items()})"; }
for rep in 1 2; do
for w in cfg3_ont_diploid_2500x60x2kbp cfg5_hifi_diploid_2500x40x2kbp cfg2_ont_diploid_500x60x2kbp; do
  JTK_MCMC_SPLIT=0 timeout 300 python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "$w general-only"
  timeout 300 python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "$w light-168   "
  JTK_LC_LIB=$L176 timeout 300 python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "$w light-176   "
done; done
