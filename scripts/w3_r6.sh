#!/bin/bash
# Diagnostic (GPU box), round 6: the general chain kernel built for three waves per SIMD (168 registers instead of 248:
# -DJTK_MCMC_WAVES=3, exp_w3) -- a general chain wave then takes one pair-HMM wave's registers instead of two.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/w3_r6.txt
echo "== $(date -u +%FT%TZ)" >> $OUT
W3=$PWD/jtk_amd/_build/exp_w3/libjtk_lc_w3.so
short() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
sp=d['roofline']['serial_pass']['kernel_ms']
print('value %.1f ms_per_step %.1f serial %s' % (d['value'], d['ms_per_step'], {k:round(v,1) for k,v in sp.items()}))
"; }
echo "w3 parity: $(JTK_LC_LIB=$W3 timeout 1500 python3 -m pytest tests/test_gpu_defining_shapes.py tests/test_gpu_parity.py -x -q -m gpu -k 'not large_pileups and not beyond_1023 and not recursive_split' 2>&1 | tail -1)" | tee -a $OUT
B="--steps 8 --warmup 2 --no-cpu-baseline --no-shard8 --no-e2e"
echo "-- cfg3 product: $(timeout 600 python3 bench.py $B 2>/dev/null | short)" | tee -a $OUT
echo "-- cfg3 w3:      $(JTK_LC_LIB=$W3 timeout 600 python3 bench.py $B 2>/dev/null | short)" | tee -a $OUT
echo "-- cfg2 product: $(timeout 600 python3 bench.py --workload cfg2_ont_diploid_500x60x2kbp $B 2>/dev/null | short)" | tee -a $OUT
echo "-- cfg2 w3:      $(JTK_LC_LIB=$W3 timeout 600 python3 bench.py --workload cfg2_ont_diploid_500x60x2kbp $B 2>/dev/null | short)" | tee -a $OUT
C="--workload cfg4_ont_4copy_2500x160x2kbp --chunks 500 --steps 1 --warmup 1 --no-cpu-baseline --no-shard8 --no-e2e"
echo "-- cfg4 (500 chunks) product: $(timeout 900 python3 bench.py $C 2>/dev/null | short)" | tee -a $OUT
echo "-- cfg4 (500 chunks) w3:      $(JTK_LC_LIB=$W3 timeout 900 python3 bench.py $C 2>/dev/null | short)" | tee -a $OUT
