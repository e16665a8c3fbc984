#!/bin/bash
# GPU box: wave-instruction counts of the chain kernel for chunks run ALONE (scripts/chain_solo.py), one rocprofv3 --pmc pass per
# counter group; the difference between two chunks of the same shape with different event counts is the cost of an event in
# instructions.  usage: chain_pmc.sh tag ids [--refit]
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
TAG=$1; IDS=$2; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES"
G2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
G3="SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32"
n=0
for G in "$G1" "$G2" "$G3"; do
  n=$((n+1))
  rm -rf $OUT/cpmc_${TAG}_g$n
  timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/cpmc_${TAG}_g$n -- python3 $REPO/scripts/chain_solo.py --ids $IDS "$@" > $OUT/cpmc_${TAG}_g$n.log 2>&1
  echo "pass $n rc=$?"
done
cd $REPO && python3 - $OUT $TAG $IDS <<'PY'
import collections, csv, glob, sys
root, tag, ids = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
rows = collections.defaultdict(dict)   # dispatch order -> counter -> value
for f in sorted(glob.glob(f"{root}/cpmc_{tag}_g*/**/*_counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "mcmc_kernel" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    # a chunk's chain runs in mcmc_kernel_light or mcmc_kernel (both are launched; one of them exits at once): add the pair
    disp = sorted(per)
    for i in range(0, len(disp) - 1, 2):
        for k, v in list(per[disp[i]].items()) :
            rows[i // 2][k] = v + per[disp[i + 1]].get(k, 0.0)
names = sorted({k for r in rows.values() for k in r})
print("chunk " + " ".join(names))
for i in sorted(rows):
    print((ids[i] if i < len(ids) else "?") + " " + " ".join("%.4g" % rows[i].get(k, float("nan")) for k in names))
PY
