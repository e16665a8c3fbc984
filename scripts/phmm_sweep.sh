#!/bin/bash
# Diagnostic (GPU box): one pair-HMM pass with parts of the kernel's HBM traffic switched off (results are garbage; timing only).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for fl in "" "-DJTK_PHMM_X_NOFLUSH" "-DJTK_PHMM_X_NOSTORE" "-DJTK_PHMM_X_NOLOAD" "-DJTK_PHMM_X_NOSTORE -DJTK_PHMM_X_NOLOAD -DJTK_PHMM_X_NOFLUSH"; do
  export JTK_LC_LIB=$(python3 -c "import jtk_amd.build as b; print(b.build_experiment('probe1', '''$fl'''))") || exit 1
  echo "[$fl]: $(timeout 200 python3 scripts/phmm_single_pass.py 2>&1 | tail -1)"
done
unset JTK_LC_LIB   # the product library was never touched
