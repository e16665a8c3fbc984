#!/bin/bash
# Diagnostic (GPU box): the chain launch as light (168 registers) + general (248) kernels against the general kernel alone
# (JTK_MCMC_SPLIT=0): parity of the chain tests, then the serial pass and the 4-in-flight throughput of both.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_defining_shapes.py -m gpu -x -q 2>&1 | tail -3
for split in 1 0; do
  echo "== JTK_MCMC_SPLIT=$split"
  JTK_MCMC_SPLIT=$split timeout 300 python3 bench.py --streams 1 --steps 2 --no-cpu-baseline --no-e2e --no-shard8 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial', round(d['value'],1), d['roofline']['serial_pass']['kernel_ms'])"
  JTK_MCMC_SPLIT=$split timeout 300 python3 bench.py --steps 12 --no-cpu-baseline --no-e2e --no-shard8 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('4 in flight', round(d['value'],1), d['ms_per_step'], d['roofline']['serial_pass']['kernel_ms'])"
done
