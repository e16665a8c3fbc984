#!/bin/bash
# GPU box, round 6: solo chain timings (one chunk per call: the chain kernel's time is that chunk's chain) of the chunks round 5
# listed -- round-5 chain (exp_chainr5), fixed blocks (product), fixed blocks without the request ahead (exp_la64).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/chain_solo_r6.txt
echo "== $(date -u +%FT%TZ)" > $OUT
for v in chainr5 product la64; do
  if [ $v = product ]; then L=""; else L=$PWD/jtk_amd/_build/exp_$v/libjtk_lc_$v.so; fi
  echo "-- $v" >> $OUT
  JTK_LC_LIB=$L timeout 600 python3 scripts/chain_pieces.py --default-model --solo 0,1,2,3,4,5,573,303,6,591 2>&1 | grep SOLO >> $OUT
done
cat $OUT
