#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 120 python3 scripts/k2_repro.py 101 10 2>&1 | grep REPRO
timeout 120 python3 scripts/k2_repro.py 102 4 2>&1 | grep REPRO
for f in K2_NO_UNI_T K2_NO_UNI_C0 K2_NO_UNI_FLAGS K2_NO_UNI_LAB; do
  export JTK_LC_LIB=$(python3 -c "import jtk_amd.build as b; print(b.build_experiment('$f', '''-D$f'''))" 2>/dev/null) || { echo build failed $f; continue; }
  echo $f; timeout 120 python3 scripts/k2_repro.py 101 10 2>&1 | grep REPRO
done
} > gpurun_out/k2_bisect.log 2>&1
cat gpurun_out/k2_bisect.log
