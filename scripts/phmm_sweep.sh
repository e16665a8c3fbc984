#!/bin/bash
# Diagnostic (GPU box): one pair-HMM pass with parts of the kernel's HBM traffic switched off (results are garbage; timing only).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for fl in "" "-DJTK_PHMM_X_NOFLUSH" "-DJTK_PHMM_X_NOSTORE" "-DJTK_PHMM_X_NOLOAD" "-DJTK_PHMM_X_NOSTORE -DJTK_PHMM_X_NOLOAD -DJTK_PHMM_X_NOFLUSH"; do
  JTK_EXTRA_HIPCC_FLAGS="$fl" python3 -c "import jtk_amd.build as b; b.build(force=True)" 2>&1 | grep -i " error"
  echo "[$fl]: $(timeout 200 python3 scripts/phmm_single_pass.py 2>&1 | tail -1)"
done
python3 -c "import jtk_amd.build as b; b.build(force=True)" 2>&1 | grep -i " error"
