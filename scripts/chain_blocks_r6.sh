#!/bin/bash
# GPU box, round 6: the diploid chain with fixed 64-position blocks (records of the next block requested a block ahead, the entries'
# look-up in flight during the logarithms): parity of everything that runs a chain, then the bench lines the chain decides.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/chain_blocks_r6.txt
echo "== $(date -u +%FT%TZ)" >> $OUT
echo "parity: $(timeout 2400 python3 -m pytest tests/test_golden.py tests/test_gpu_defining_shapes.py tests/test_gpu_parity.py tests/test_gpu_shapes.py -x -q -m gpu 2>&1 | tail -1)" | tee -a $OUT
short() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
sp=d['roofline']['serial_pass']['kernel_ms']
pr=d['per_rank'][0]
print('value %.1f ms_per_step %.1f serial %s slowest chain %.1f median %.1f' % (d['value'], d['ms_per_step'], {k:round(v,1) for k,v in sp.items()}, pr['slowest_chunk_chain_ms'] or 0, pr['median_chain_ms'] or 0))
"; }
B="--steps 8 --warmup 2 --no-cpu-baseline --no-shard8 --no-e2e"
echo "-- cfg3: $(timeout 600 python3 bench.py $B 2>/dev/null | short)" | tee -a $OUT
echo "-- cfg2: $(timeout 600 python3 bench.py --workload cfg2_ont_diploid_500x60x2kbp $B 2>/dev/null | short)" | tee -a $OUT
echo "-- cfg5: $(timeout 600 python3 bench.py --workload cfg5_hifi_diploid_2500x40x2kbp $B 2>/dev/null | short)" | tee -a $OUT
echo "-- cfg3 again: $(timeout 600 python3 bench.py $B 2>/dev/null | short)" | tee -a $OUT
