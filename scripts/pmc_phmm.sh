#!/bin/bash
# Runs on the GPU box (via gpurun): SQ / SQC issue, stall and LDS counters of the pair-HMM kernels, one rocprofv3 --pmc
# pass per group of <= 8 SQ counters (MI355X_MICROARCH.md "rocprofv3 PMC slots"), on ONE pair-HMM pass over a
# 500-chunk cfg 2 batch (scripts/phmm_single_pass.py).  scripts/summarize_pmc.py condenses the CSVs for profiles/.
# usage: pmc_phmm.sh [tag] [n_chunks]
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
TAG=${1:-r06}
NCH=${2:-500}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/pmc_counters_avail.txt 2>&1
have() { grep -qw "$1" $OUT/pmc_counters_avail.txt; }
pick() { local o=""; for c in "$@"; do have $c && o="$o $c"; done; echo $o; }
G1=$(pick SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR)
G2=$(pick SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM)
G3=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU)
G4=$(pick SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_FLAT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64)
G5=$(pick SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_WAVES_EQ_64 SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_CYCLES)
G6=$(pick GRBM_GUI_ACTIVE GRBM_COUNT TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum)
echo "python3 scripts/phmm_single_pass.py $NCH" > $OUT/pmc_cmd_$TAG.txt
sha256sum $REPO/jtk_amd/_build/libjtk_lc.so | cut -c1-16 > $OUT/pmc_libsha_$TAG.txt
n=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6"; do
  n=$((n+1))
  [ -z "$G" ] && continue
  rm -rf $OUT/pmc_${TAG}_g$n
  echo "pass $n: $G"
  timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc_${TAG}_g$n -- python3 $REPO/scripts/phmm_single_pass.py $NCH > $OUT/pmc_${TAG}_g$n.log 2>&1
  echo "  rc=$?"
done
rm -rf $OUT/pmc_${TAG}_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pmc_${TAG}_stats -- python3 $REPO/scripts/phmm_single_pass.py $NCH > $OUT/pmc_${TAG}_stats.log 2>&1
echo "stats rc=$?"
cd $REPO && python3 scripts/summarize_pmc.py $OUT $TAG > $OUT/pmc_summary_$TAG.txt 2>&1
cat $OUT/pmc_summary_$TAG.txt | tail -60
