#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "device_side_validation or failed_chunk_inside or one_shot_slicing" 2>&1 | tail -3 | tee gpurun_out/newtests_r6.txt
PMC_PHMM=1 bash scripts/final_round.sh r06 > gpurun_out/final_round_r06.log 2>&1
tail -8 gpurun_out/final_round_r06.log
timeout 2400 python3 -m pytest tests -q -m gpu > gpurun_out/gputest_r6_final.txt 2>&1
tail -4 gpurun_out/gputest_r6_final.txt
