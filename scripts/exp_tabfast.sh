#!/bin/bash
# GPU box: parity of every table-driven K-way chain case of the suite, then the cfg-4 and headline bench lines.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout 900 python3 -m pytest tests -q -m gpu -x --durations=15 -k "cfg4 or chain or copy_numbers or campaign or recursive or large_pileups or beyond_511 or cluster_features or size_only or sweep or heterogeneous" > $O/tabfast_tests.log 2>&1
echo "tests rc=$?"; tail -25 $O/tabfast_tests.log
timeout 600 python3 bench.py --workload cfg4_ont_4copy_2500x160x2kbp --steps 1 --warmup 0 --no-e2e --no-cpu-baseline --no-shard8 > $O/tabfast_cfg4.json 2> $O/tabfast_cfg4.err

for f in cfg4; do python3 -c "import json,sys; d=json.loads(open('$O/tabfast_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],1), {k: round(v) for k, v in d['roofline']['serial_pass']['kernel_ms'].items()})" 2>&1 | tail -1; done
