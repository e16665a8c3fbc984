#!/bin/bash
# GPU box, end of round 6: everything profiles/r06_* holds, on the library that carries jtk_lc_session_trace (the product kernels'
# machine code is unchanged, profiles/r06_trace_isa_same.txt), then the whole GPU suite on the same library.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
PMC_PHMM=1 bash scripts/final_round.sh r06 > gpurun_out/final_round_r06.log 2>&1
tail -8 gpurun_out/final_round_r06.log
timeout 1500 python3 -m pytest tests -q -m gpu > gpurun_out/gputest_r6_final.txt 2>&1
tail -4 gpurun_out/gputest_r6_final.txt
sha256sum jtk_amd/_build/libjtk_lc.so | cut -c1-16
