#!/bin/bash
# Diagnostic (GPU box): rebuild with per-chunk chain counters, run one bench pass, keep the slowest chunks.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
JTK_EXTRA_HIPCC_FLAGS="-DJTK_MCMC_STATS" python3 -c "import jtk_amd.build as b; b.build(force=True)" 2>&1 | grep -i error
python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/stats_raw.txt 2>&1
grep K2STAT gpurun_out/stats_raw.txt | sort -k8 -n -r | head -12
grep K2STAT gpurun_out/stats_raw.txt | awk '{c+=$8; w+=$10; wi+=$12; e+=$14; r+=$16; s+=$18; nw+=$20; ne+=$22; na+=$24; nc+=$26} END {print "TOTAL cyc",c,"walk",w,"win",wi,"event",e,"rebuild",r,"steps",s,"windows",nw,"events",ne,"accepts",na,"changed",nc}'
python3 -c "import jtk_amd.build as b; b.build(force=True)" 2>&1 | grep -i error
