#!/bin/bash
# Diagnostic (GPU box): parity of the replaying phmm_kernel (modification tables + polished chunks against the oracle), then one
# pair-HMM pass over 30,000 reads with the product and with -DJTK_PHMM_REPLAY=0 (jtk_amd/_build/exp_noreplay, built beforehand).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "modification_table or cluster_polished or cluster_chunks_matches" 2>&1 | tail -5
for rep in 1 2; do
echo "replay:   $(timeout 200 python3 scripts/phmm_single_pass.py 500 2>/dev/null | tail -1)"
echo "noreplay: $(JTK_LC_LIB=$PWD/jtk_amd/_build/exp_noreplay/libjtk_lc_noreplay.so timeout 200 python3 scripts/phmm_single_pass.py 500 2>/dev/null | tail -1)"
done
