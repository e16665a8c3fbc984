#!/bin/bash
# (rounds 1-4: the ring size was the compile-time JTK_MCMC_SEG_LOG; since round 5 it is a launch parameter -- JTK_SEG_LOG_LIGHT /
# JTK_SEG_LOG_GENERAL in mcmc_kernels.hip, one jump table per length -- and this probe needs those two edited instead)
# Diagnostic (GPU box): the chain kernel's ring size (JTK_MCMC_SEG_LOG: 5 = 4096 draws / 70 KB of LDS per workgroup,
# 4 = 2048 draws / 24 KB of ring, 3 = 1024 draws / 12 KB) against parity, the serial step and the 4-in-flight throughput.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
for seg in ${SEGS:-3 4}; do
  export JTK_LC_LIB=$(python3 -c "import jtk_amd.build as b; print(b.build_experiment('probe1', '''-DJTK_MCMC_SEG_LOG=$seg'''))") || exit 1
  echo "== SEG_LOG $seg"
  if [ $seg != 4 ]; then timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain or size_only or features or full_size" 2>&1 | tail -2; fi
  timeout 300 python3 bench.py --streams 1 --steps 2 --no-cpu-baseline --no-e2e --no-shard8 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial', round(d['value'],1), d['roofline']['serial_pass'])"
  timeout 300 python3 bench.py --steps 12 --no-cpu-baseline --no-e2e --no-shard8 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('4 in flight', round(d['value'],1), d['roofline']['serial_pass'])"
done
unset JTK_LC_LIB   # the product library was never touched
