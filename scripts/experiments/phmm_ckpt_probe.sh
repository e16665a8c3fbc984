#!/bin/bash
# Runs on the GPU box: one pair-HMM pass (finalize included) over 30,000 reads with the product library and with the TIMING-ONLY
# checkpoint emulations built by phmm_ckpt_probe.py (they travel with the snapshot).  See that file for what each variant is.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
echo "product: $(timeout 200 python3 scripts/phmm_single_pass.py 500 2>/dev/null | tail -1)"
for w in ckpt4 ckpt4_fwd2 ckpt8 ckpt8_fwd2 fwd2 nostream; do
  echo "$w: $(JTK_LC_LIB=$PWD/jtk_amd/_build/exp_ckpt_$w/libjtk_lc_$w.so timeout 200 python3 scripts/phmm_single_pass.py 500 2>/dev/null | tail -1)"
done; done
