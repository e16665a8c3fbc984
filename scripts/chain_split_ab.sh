#!/bin/bash
# Diagnostic (GPU box): light + general chain kernels against the general kernel alone, on all three diploid workloads
# (12 queues, no null-stream memset).  cfg 3 / 5 / 2, same box.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k: round(v) for k, v in d['roofline']['serial_pass']['kernel_ms'].items()})"; }
for rep in 1 2; do
for w in cfg3_ont_diploid_2500x60x2kbp cfg5_hifi_diploid_2500x40x2kbp cfg2_ont_diploid_500x60x2kbp; do
  JTK_MCMC_SPLIT=0 timeout 300 python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "$w general-only"
  timeout 300 python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "$w split       "
done; done
GPU_MAX_HW_QUEUES=8 timeout 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "cfg3 split, 8 queues"
