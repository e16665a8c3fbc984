#!/bin/bash
# GPU box, round 6: the diploid chain after the fixed-block windows -- per-proposal / per-event cycles (statistics build, default
# model, 600 chunks), the producer's own counters, and solo timings of the chunks round 5 listed (product library).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/chain_pieces_r6.txt
echo "== $(date -u +%FT%TZ)" > $OUT
JTK_LC_LIB=$PWD/jtk_amd/_build/exp_k2stats/libjtk_lc_k2stats.so timeout 900 python3 scripts/chain_pieces.py --chunks 600 --default-model > gpurun_out/chain_pieces_raw.txt 2>&1
grep -v "^K2STAT\|^K2PROD\|^K2WAIT" gpurun_out/chain_pieces_raw.txt | tail -40 >> $OUT
echo "-- producer counters (first 5 workgroups): " >> $OUT
grep "^K2PROD" gpurun_out/chain_pieces_raw.txt | head -5 >> $OUT
python3 - >> $OUT <<'PY'
import re
g=p=j=w=0; n=0
for line in open("gpurun_out/chain_pieces_raw.txt"):
    m=re.match(r"K2PROD wr (\d+) sleeps (\d+) cyc_gen (\d+) cyc_parse (\d+) cyc_jump (\d+)", line)
    if m:
        wr,sl,cg,cp,cj=map(int,m.groups()); w+=wr; g+=cg; p+=cp; j+=cj; n+=1
if n: print("producers %d: per draw gen %.1f parse %.1f jump %.1f cycles" % (n, g/w, p/w, j/w))
PY
echo "-- solo (product library, default model)" >> $OUT
timeout 600 python3 scripts/chain_pieces.py --default-model --solo 0,4,5,573,303,6,591 2>&1 | grep SOLO >> $OUT
cat $OUT
rm -f gpurun_out/chain_pieces_raw.txt
