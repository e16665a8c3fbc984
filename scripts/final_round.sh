#!/bin/bash
# Runs on the GPU box (via gpurun): everything profiles/ holds for a round's final build -- rocprofv3 stats + HBM traffic of the
# default bench command, the pair-HMM issue counters, the other BASELINE configurations, the default bench line.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
O=gpurun_out
if [ "${PROFILE:-1}" = "1" ]; then   # PROFILE=0: the profiles were taken already on these sources
bash scripts/profile_bench.sh $TAG > $O/final_profile_$TAG.log 2>&1
# (PMC_PHMM=1: the pair-HMM issue counters as well -- six more rocprofv3 passes; phmm_sweep.hip / phmm_pair.hip / phmm_wide.hip did
# not change in round 5, profiles/r04_pmc_issue_phmm.txt describes the kernels as they are)
[ "${PMC_PHMM:-0}" = "1" ] && bash scripts/pmc_phmm.sh $TAG > $O/final_pmc_$TAG.log 2>&1
fi
cd ${GRAFT_REPO_ROOT:-/root/repo}
# the bench lines below carry roofline.traffic from THIS profile (bench.py matches it by the hash of the kernel sources)
[ -s $O/prof_traffic_$TAG.json ] && cp $O/prof_traffic_$TAG.json profiles/${TAG}_pmc_traffic.json
# (the default line first: it is the one the driver repeats; the other configurations after it)
timeout 1200 python3 bench.py --steps 20 --warmup 5 > $O/bench_${TAG}_final.json 2> $O/bench_${TAG}_final.err   # the driver's command
# (cfg 4 WITH its cpu_baseline rung: the oracle on the first chunks of the same workload, a few minutes of host time)
timeout 2700 python3 bench.py --workload cfg4_ont_4copy_2500x160x2kbp --steps 1 --warmup 0 --no-e2e --no-shard8 > $O/bench_${TAG}_cfg4_2500.json 2> $O/bench_${TAG}_cfg4_2500.err
timeout 600 python3 bench.py --workload cfg5_hifi_diploid_2500x40x2kbp --steps 8 --warmup 2 --no-e2e --no-cpu-baseline --no-shard8 > $O/bench_${TAG}_cfg5.json 2> $O/bench_${TAG}_cfg5.err
timeout 600 python3 bench.py --workload cfg2_ont_diploid_500x60x2kbp --steps 8 --warmup 2 --no-e2e --no-cpu-baseline --no-shard8 > $O/bench_${TAG}_cfg2.json 2> $O/bench_${TAG}_cfg2.err
timeout 600 python3 scripts/poisson_coverage_bench.py 500 > $O/poisson_$TAG.log 2>&1
# One library, one set of numbers: a bench line whose lib_sha16 is not the hash of the library the committed rocprof / PMC profile was
# taken on (gpurun_out/prof_libsha_$TAG.txt, written by profile_bench.sh) is set aside as .STALE -- it must not reach profiles/.
python3 - "$TAG" <<'PY'
import json, os, sys
tag = sys.argv[1]
O = "gpurun_out"
try:
    want = open(f"{O}/prof_libsha_{tag}.txt").read().strip()
except OSError:
    want = None
for f in ("cfg4_2500", "cfg5", "cfg2", "final"):
    path = f"{O}/bench_{tag}_{f}.json"
    try:
        got = json.loads(open(path).read().strip().splitlines()[-1]).get("lib_sha16")
    except (OSError, ValueError, IndexError):
        print("final_round:", path, "is missing or not a bench line")
        continue
    if want is None or got != want:
        os.replace(path, path + ".STALE")
        print(f"final_round: {path} was measured on library {got}, the profile on {want}: moved to {path}.STALE")
    else:
        print(f"final_round: {path} lib_sha16 {got} == profile's")
PY
# summary: the last line of every bench file that survived the check (a file set aside as .STALE is named, not parsed); the script
# FAILS when the default line -- the one the driver repeats -- is not there
python3 - "$TAG" <<'PY'
import json, os, sys
tag = sys.argv[1]
O = "gpurun_out"
bad = False
for f in ("cfg4_2500", "cfg5", "cfg2", "final"):
    path = f"{O}/bench_{tag}_{f}.json"
    if not os.path.exists(path):
        print(f, "-- no line", "(set aside as .STALE)" if os.path.exists(path + ".STALE") else "(missing)")
        bad = bad or f == "final"
        continue
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), round(d["ms_per_step"], 1), {k: round(v) for k, v in d["roofline"]["serial_pass"]["kernel_ms"].items()})
    except (ValueError, IndexError, KeyError) as e:
        print(f, "-- unreadable:", e)
        bad = bad or f == "final"
if bad:
    print("final_round: FAILED -- no valid default bench line for", tag)
    sys.exit(1)
PY
RC=$?
tail -2 $O/poisson_$TAG.log
exit $RC
