#!/bin/bash
# Diagnostic (GPU box), round 6: should phmm_pair_kernel replay its forward sweep instead of streaming pairs through HBM
# (missing 3)?  Timing-only builds of phmm_pair.hip (scripts/build_variant.py, built beforehand; their tables
# are garbage) against the product, one pair-HMM pass over 1,000 HiFi pile-ups = 40,000 reads (scripts/phmm_single_pass.py):
#   x_pair_nostream   -DJTK_PAIR_X_NOSTREAM    no pair is stored or loaded: the bound on what ANY replay can win
#   x_pair_replaycost -DJTK_PAIR_X_REPLAYCOST  a quarter of the rows stored / loaded (the checkpoints' share) + one forward step
#                                              per diagonal of the sweep back: what a replay would cost
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/pair_probe_r6.txt
echo "== $(date -u +%FT%TZ) one pair-HMM pass, 1,000 hifi_diploid pile-ups (40,000 reads), kernel_ms" > $OUT
one() {  # name, lib ("" = product)
  for rep in 1 2; do
    r=$(JTK_LC_LIB=$2 timeout 300 python3 scripts/phmm_single_pass.py 1000 hifi_diploid 2>/dev/null | tail -1)
    echo "$1: $r" | tee -a $OUT
  done
}
one product ""
for v in x_pair_nostream x_pair_replaycost; do
  lib=$PWD/jtk_amd/_build/exp_$v/libjtk_lc_$v.so
  [ -f $lib ] || { echo "$v: no library" | tee -a $OUT; continue; }
  one $v $lib
done
one product ""
