#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + HBM-traffic counters for a bench command.
# Summaries land in gpurun_out/prof_*; scripts/summarize_prof.py condenses them for profiles/.
# usage: profile_bench.sh [tag] [bench args...]      (default: the default bench command, serial slices)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
TAG=${1:-r06}
shift || true
# --no-shard8: every launch in the profile is a launch of the FULL workload (round 3 left the 313-chunk shard runs in: their
# small launches diluted the per-launch averages by 1.4x)
ARGS="${@:---steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-shard8}"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_stats_$TAG $OUT/prof_fetch_$TAG $OUT/prof_write_$TAG
echo "python3 bench.py $ARGS" > $OUT/prof_cmd_$TAG.txt
sha256sum $REPO/jtk_amd/_build/libjtk_lc.so | cut -c1-16 > $OUT/prof_libsha_$TAG.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_$TAG -- python3 $REPO/bench.py $ARGS > $OUT/prof_stats_$TAG.log 2>&1
echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_$TAG -- python3 $REPO/bench.py $ARGS > $OUT/prof_fetch_$TAG.log 2>&1
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_$TAG -- python3 $REPO/bench.py $ARGS > $OUT/prof_write_$TAG.log 2>&1
echo "write rc=$?"
cd $REPO && python3 scripts/summarize_prof.py $OUT $TAG > $OUT/prof_summary_$TAG.txt 2>&1
tail -5 $OUT/prof_summary_$TAG.txt
