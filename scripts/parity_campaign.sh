#!/bin/bash
# Runs on the GPU box: randomized parity against the CPU oracle beyond the committed tests (random chunk ids = RNG streams,
# random shapes), then the default bench line.  usage: parity_campaign.sh [seed]
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-31}
timeout 900 python3 scripts/parity_sweep_full.py 60 $S 2>&1 | tail -3
timeout 600 python3 scripts/parity_headline.py 256 $S 2>&1 | tail -2
timeout 600 python3 scripts/parity_headline.py 160 $((S+1)) hifi_diploid 2>&1 | tail -2
timeout 900 python3 scripts/parity_headline.py 12 $((S+2)) ont_4copy 2>&1 | tail -2
timeout 600 python3 scripts/parity_sweep.py 40 $S 2>&1 | tail -2
