#!/bin/bash
# Runs on the GPU box: one pair-HMM pass over 30,000 reads with the product library and with the non-temporal experiment
# libraries built by phmm_nt_stream.py (they travel with the snapshot), then the parity of the "both" variant.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
echo "product: $(timeout 120 python3 scripts/phmm_single_pass.py 500 2>/dev/null | tail -1)"
for w in stores loads both; do
  echo "nt $w: $(JTK_LC_LIB=$PWD/jtk_amd/_build/exp_nt_$w/libjtk_lc_nt_$w.so timeout 120 python3 scripts/phmm_single_pass.py 500 2>/dev/null | tail -1)"
done; done
JTK_LC_LIB=$PWD/jtk_amd/_build/exp_nt_both/libjtk_lc_nt_both.so timeout 200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "table or golden or full_size" 2>&1 | tail -1
