#!/bin/bash
# Diagnostic (GPU box), round 6: one pair-HMM pass over 30,000 reads (scripts/phmm_single_pass.py 500) with the product and with
# variant builds of phmm_sweep.hip (scripts/build_variant.py, built beforehand).  Variants whose name starts with x_ are
# timing-only (their tables are garbage); the others are candidates and get the table parity tests first.
#   usage: scripts/phmm_probe_r6.sh <variant> ...
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/phmm_probe_r6.txt
echo "== $(date -u +%FT%TZ) variants: $*" >> $OUT
one() {  # name, lib ("" = product)
  for rep in 1 2; do
    if [ -z "$2" ]; then r=$(timeout 300 python3 scripts/phmm_single_pass.py 500 2>/dev/null | tail -1)
    else r=$(JTK_LC_LIB=$2 timeout 300 python3 scripts/phmm_single_pass.py 500 2>/dev/null | tail -1); fi
    echo "$1: $r" | tee -a $OUT
  done
}
one product ""
for v in "$@"; do
  lib=$PWD/jtk_amd/_build/exp_$v/libjtk_lc_$v.so
  [ -f $lib ] || { echo "$v: no library" | tee -a $OUT; continue; }
  case $v in
    x_*) ;;
    *) echo "$v parity: $(JTK_LC_LIB=$lib timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k 'modification_table or cluster_polished or cluster_chunks_matches' 2>&1 | tail -1)" | tee -a $OUT ;;
  esac
  one $v $lib
done
one product ""
