#!/bin/bash
# Diagnostic (GPU box): what the acquire / release fences of the shared stripe set cost (JTK_STRIPE_UNFENCED=1 skips them:
# measurement only, a stripe may then be read stale after moving between XCDs).  Same box, alternating.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for uf in 0 1; do
  if [ $uf = 1 ]; then export JTK_STRIPE_UNFENCED=1; else unset JTK_STRIPE_UNFENCED; fi
  timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('unfenced $uf:', round(d['value'],1), round(d['ms_per_step'],1), {k: round(v) for k, v in d['roofline']['serial_pass']['kernel_ms'].items()})"
done
done
