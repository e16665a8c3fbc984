#!/bin/bash
# GPU box, round 6: the variant filter straight from the row sums (column_filter_fused_kernel; no table written): whole GPU suite
# minus the slowest single-chain cases, then the bench line with the fused filter and with JTK_FILTER_FUSED=0.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/fused_filter_r6.txt
echo "== $(date -u +%FT%TZ)" > $OUT
echo "parity (fused): $(timeout 2400 python3 -m pytest tests -x -q -m gpu -k 'not test_bench_gpu and not beyond_1023 and not large_pileups' 2>&1 | tail -1)" | tee -a $OUT
short() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
sp=d['roofline']['serial_pass']['kernel_ms']
print('value %.1f ms_per_step %.1f serial %s' % (d['value'], d['ms_per_step'], {k:round(v,1) for k,v in sp.items()}))
"; }
B="--steps 10 --warmup 2 --no-cpu-baseline --no-shard8 --no-e2e"
echo "-- fused:   $(timeout 600 python3 bench.py $B 2>/dev/null | short)" | tee -a $OUT
echo "-- unfused: $(JTK_FILTER_FUSED=0 timeout 600 python3 bench.py $B 2>/dev/null | short)" | tee -a $OUT
echo "-- fused:   $(timeout 600 python3 bench.py $B 2>/dev/null | short)" | tee -a $OUT
echo "-- unfused: $(JTK_FILTER_FUSED=0 timeout 600 python3 bench.py $B 2>/dev/null | short)" | tee -a $OUT
