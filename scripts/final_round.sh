#!/bin/bash
# Runs on the GPU box (via gpurun): everything profiles/ holds for a round's final build -- rocprofv3 stats + HBM traffic of the
# default bench command, the pair-HMM issue counters, the other BASELINE configurations, the default bench line.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
O=gpurun_out
if [ "${PROFILE:-1}" = "1" ]; then   # PROFILE=0: the profiles were taken already on these sources
bash scripts/profile_bench.sh $TAG > $O/final_profile_$TAG.log 2>&1
bash scripts/pmc_phmm.sh $TAG > $O/final_pmc_$TAG.log 2>&1
fi
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python3 bench.py --workload cfg4_ont_4copy_2500x160x2kbp --steps 1 --warmup 0 --no-e2e --no-cpu-baseline --no-shard8 > $O/bench_${TAG}_cfg4_2500.json 2> $O/bench_${TAG}_cfg4_2500.err
timeout 600 python3 bench.py --workload cfg5_hifi_diploid_2500x40x2kbp --steps 8 --warmup 2 --no-e2e --no-cpu-baseline --no-shard8 > $O/bench_${TAG}_cfg5.json 2> $O/bench_${TAG}_cfg5.err
timeout 600 python3 bench.py --workload cfg2_ont_diploid_500x60x2kbp --steps 8 --warmup 2 --no-e2e --no-cpu-baseline --no-shard8 > $O/bench_${TAG}_cfg2.json 2> $O/bench_${TAG}_cfg2.err
timeout 600 python3 scripts/poisson_coverage_bench.py 500 > $O/poisson_$TAG.log 2>&1
timeout 1200 python3 bench.py --steps 20 --warmup 5 > $O/bench_${TAG}_final.json 2> $O/bench_${TAG}_final.err   # the driver's command
for f in cfg4_2500 cfg5 cfg2 final; do python3 -c "import json,sys; d=json.load(open('$O/bench_${TAG}_$f.json')); print('$f', round(d['value'],1), round(d['ms_per_step'],1), {k: round(v) for k, v in d['roofline']['serial_pass']['kernel_ms'].items()})" 2>&1 | tail -1; done
tail -2 $O/poisson_$TAG.log
