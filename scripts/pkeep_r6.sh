#!/bin/bash
# GPU box, round 6: records kept per parse round of the diploid chain's producer (PKEEP 48 = 16 draws of look-ahead for the last
# kept start; 56 / 58 = 8 / 6: more starts fall back to scalar draws, every parse round yields more records).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/pkeep_r6.txt
echo "== $(date -u +%FT%TZ)" > $OUT
for v in product pkeep56 pkeep58; do
  if [ $v = product ]; then L=""; else L=$PWD/jtk_amd/_build/exp_$v/libjtk_lc_$v.so; fi
  echo "-- $v" >> $OUT
  JTK_LC_LIB=$L timeout 600 python3 scripts/chain_pieces.py --default-model --solo 0,4,5,573,6,591 2>&1 | grep SOLO >> $OUT
done
for v in pkeep56 pkeep58; do
  L=$PWD/jtk_amd/_build/exp_$v/libjtk_lc_$v.so
  echo "$v parity: $(JTK_LC_LIB=$L timeout 1200 python3 -m pytest tests/test_golden.py tests/test_gpu_defining_shapes.py -x -q -m gpu 2>&1 | tail -1)" >> $OUT
done
cat $OUT
