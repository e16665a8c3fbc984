#!/bin/bash
# Diagnostic (GPU box), round 6, second pass: why does the chain's presence cost the headline step 260 ms (overlap_probe_r6:
# 627 ms per step without the chain kernels, 890 with)?  (1) parity of the current build; (2) a kernel trace of three steps
# (start / end of every dispatch: who is resident when); (3) the step with fewer persistent pair-HMM waves per CU, with the general
# chain kernel on the main stream, and with the chain's kernels confined to n CUs (JTK_CHAIN_CUS).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/overlap_probe2_r6.txt
echo "== $(date -u +%FT%TZ)" >> $OUT
echo "-- parity" | tee -a $OUT
timeout 1200 python3 -m pytest tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_defining_shapes.py -x -q -m gpu -k "not large_pileups and not beyond_1023 and not recursive_split" 2>&1 | tail -3 | tee -a $OUT
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard8 --no-e2e"
short() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
sp=d['roofline']['serial_pass']['kernel_ms']
print('value %.1f ms_per_step %.1f serial %s' % (d['value'], d['ms_per_step'], {k:round(v,1) for k,v in sp.items()}))
"; }
echo "-- default: $(timeout 600 $B 2>/dev/null | short)" | tee -a $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/trace_r6 -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shard8 --no-e2e > /dev/null 2>&1 )
python3 scripts/trace_timeline.py gpurun_out/trace_r6 2>&1 | tee -a $OUT
rm -rf gpurun_out/trace_r6
for w in 11 10 8; do echo "-- phmm waves per CU $w: $(JTK_PHMM_WAVES_PER_CU=$w timeout 600 $B 2>/dev/null | short)" | tee -a $OUT; done
echo "-- general chain kernel on the main stream: $(JTK_MCMC_SIDE=0 timeout 600 $B 2>/dev/null | short)" | tee -a $OUT
for n in 64 96 128; do echo "-- chain on $n CUs (spread): $(GPU_MAX_HW_QUEUES=24 JTK_CHAIN_CUS=$n timeout 600 $B 2>/dev/null | short)" | tee -a $OUT; done
echo "-- chain on the first 64 CUs: $(GPU_MAX_HW_QUEUES=24 JTK_CHAIN_CUS=64 JTK_CHAIN_CUS_SPREAD=0 timeout 600 $B 2>/dev/null | short)" | tee -a $OUT
echo "-- e2e: $(timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shard8 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in (d.get('e2e') or {}).items() if k in ('chunks_per_s','seconds','first_call_seconds','h2d_ms','d2h_ms','matches_resident')})
print('stage_e2e', {k:(round(v,1) if isinstance(v,float) else v) for k,v in ((d.get('stage_e2e') or {}).get('warm') or {}).items() if k.endswith('_ms')}, (d.get('stage_e2e') or {}).get('chunks_per_s_warm'))
")" | tee -a $OUT
