#!/bin/bash
# Diagnostic (GPU box): chain parity (feature-level sweep + the chain tests) and the chain kernel's time on one slice of
# the headline workload: chain_check.sh [n_seeds] [name:@other_lib.so ...]
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-4}; shift || true
if [ "$N" != "0" ]; then
python3 scripts/parity_sweep.py $N 100 2>&1 | tail -4
timeout 1200 python3 -m pytest -x -q tests/test_gpu_parity.py -k "chain_variants or size_only or cluster_features or edge_cases" tests/test_gpu_shapes.py::test_random_chain_sweep_matches_oracle 2>&1 | tail -3
fi
bash scripts/chain_time_variants.sh "product:" "$@"
