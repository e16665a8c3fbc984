#!/bin/bash
# Diagnostic (GPU box): slices in flight (each two streams; bench.py asks for 2 x streams + 4 hardware queues), headline workload.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],1), {k: round(v) for k, v in d['roofline']['serial_pass']['kernel_ms'].items()})"; }
for st in ${STREAMS:-4 5 6 8 3 4}; do
  timeout 300 python3 bench.py --streams $st --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-shard8 | line "streams $st"
done
