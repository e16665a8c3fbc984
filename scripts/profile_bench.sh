#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + HBM-traffic counters for the bench command.
# Summaries land in gpurun_out/prof_*; copy what should be judged into profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 16 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $CMD > $OUT/prof_stats.log 2>&1
echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- $CMD > $OUT/prof_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- $CMD > $OUT/prof_write.log 2>&1
echo "write rc=$?"
find $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write -name "*.csv" | head -20
