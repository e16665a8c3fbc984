#!/bin/bash
# Diagnostic (GPU box), round 6: (1) what host buffers cost to move (scripts/pcie_probe.hip); (2) parity of the device-side
# encode / gather (the committed golden + the one-shot tests); (3) the headline step with every output fetched, with results only,
# without the chain kernels (JTK_X_NOCHAIN: what sharing the CUs with the chain costs the pair-HMM family), with the light chain's
# ring at 12 KiB (exp_seg3) and with the v_cmpx row sums (exp_cmpx).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/overlap_probe_r6.txt
echo "== $(date -u +%FT%TZ)" >> $OUT
timeout 120 scripts/bin/pcie_probe 2>&1 | tee -a $OUT
echo "-- parity" | tee -a $OUT
timeout 900 python3 -m pytest tests/test_golden.py tests/test_gpu_parity.py -x -q -m gpu -k "not large_pileups and not beyond_1023 and not recursive_split" 2>&1 | tail -3 | tee -a $OUT
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard8"
short() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
sp=d['roofline']['serial_pass']['kernel_ms']
print('value %.1f ms_per_step %.1f serial %s e2e %s' % (d['value'], d['ms_per_step'], {k:round(v,1) for k,v in sp.items()}, {k:(round(v,3) if isinstance(v,float) else v) for k,v in (d.get('e2e') or {}).items() if k in ('chunks_per_s','seconds','first_call_seconds','h2d_ms','d2h_ms','matches_resident')}))
print('stage_e2e', {k:(round(v,1) if isinstance(v,float) else v) for k,v in ((d.get('stage_e2e') or {}).get('warm') or {}).items() if k.endswith('_ms')}, (d.get('stage_e2e') or {}).get('chunks_per_s_warm'))
"; }
echo "-- fetch all + e2e: $(timeout 900 $B 2>/dev/null | short)" | tee -a $OUT
echo "-- fetch results: $(timeout 600 $B --no-e2e --fetch results 2>/dev/null | short)" | tee -a $OUT
echo "-- no chain (timing only): $(JTK_X_NOCHAIN=1 timeout 600 $B --no-e2e 2>/dev/null | short)" | tee -a $OUT
echo "-- light ring 12 KiB: $(JTK_LC_LIB=$PWD/jtk_amd/_build/exp_seg3/libjtk_lc_seg3.so timeout 600 $B --no-e2e 2>/dev/null | short)" | tee -a $OUT
echo "-- cmpx: $(JTK_LC_LIB=$PWD/jtk_amd/_build/exp_cmpx/libjtk_lc_cmpx.so timeout 600 $B --no-e2e 2>/dev/null | short)" | tee -a $OUT
echo "-- fetch all again: $(timeout 600 $B --no-e2e 2>/dev/null | short)" | tee -a $OUT
