#!/bin/bash
# Diagnostic (GPU box): light + general chain kernels against the general kernel alone (JTK_MCMC_SPLIT=0) on cfg 5 and cfg 2.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k: round(v) for k, v in d['roofline']['serial_pass']['kernel_ms'].items()})"; }
for rep in 1 2; do
for w in cfg5_hifi_diploid_2500x40x2kbp cfg2_ont_diploid_500x60x2kbp; do
for split in 1 0; do
  JTK_MCMC_SPLIT=$split timeout 300 python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "$w split=$split"
done; done; done
