#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() {
  JTK_EXTRA_HIPCC_FLAGS="$1" python3 -c "import jtk_amd.build as b; b.build(force=True)" 2>&1 | grep -i " error"
  echo "=== [$1] $2"
  env $2 timeout 200 python3 scripts/phmm_debug.py 2000 3 2>&1 | grep "differing\|by table" | head -80
}
run "" "X=1"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -5; timeout 200 python scripts/phmm_single_pass.py 500 2>&1 | tail -2
