#!/bin/bash
# GPU box: issue / stall counters of the chain kernel on ONE full-size 4-copy pile-up (mcmc_chain_tab at K = 2, 3, 4 on 160 reads):
# is the K-way chain short of instructions or short of issue?  usage: tab_pmc.sh tag
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
TAG=${1:-tab}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES"
G2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
G3="SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_VMEM"
n=0
for G in "$G1" "$G2" "$G3"; do
  n=$((n+1))
  rm -rf $OUT/tpmc_${TAG}_g$n
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/tpmc_${TAG}_g$n -- python3 $REPO/scripts/experiments/tab_event/chain_ms.py 1 > $OUT/tpmc_${TAG}_g$n.log 2>&1
  echo "pass $n rc=$?"
done
cd $REPO && python3 - $OUT $TAG <<'PY'
import collections, csv, glob, sys
root, tag = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float)
for f in sorted(glob.glob(f"{root}/tpmc_{tag}_g*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "mcmc_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot):
    print("%-24s %.4g" % (k, tot[k]))
wc = tot.get("SQ_WAVE_CYCLES", 0)
if wc:
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_VMEM"):
        if k in tot:
            print("%-36s %.3f" % (k + " / SQ_WAVE_CYCLES", tot[k] / wc))
    ins = sum(tot.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
    print("wave-instructions %.4g; per wave quad-cycle %.3f" % (ins, ins / wc))
PY
