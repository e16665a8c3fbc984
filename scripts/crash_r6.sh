#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/crash_r6.txt
echo "== fused" > $OUT
timeout 1500 python3 -m pytest tests -x -v -m gpu -k 'not test_bench_gpu and not beyond_1023 and not large_pileups' 2>&1 | tail -25 >> $OUT
echo "== unfused" >> $OUT
JTK_FILTER_FUSED=0 timeout 1500 python3 -m pytest tests -x -q -m gpu -k 'not test_bench_gpu and not beyond_1023 and not large_pileups' 2>&1 | tail -5 >> $OUT
cat $OUT
