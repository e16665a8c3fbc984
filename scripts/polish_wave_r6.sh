#!/bin/bash
# GPU box, round 6: band_prep and rethread as wave-per-read kernels: the whole GPU suite (every full-path test compares the
# re-threaded ops byte for byte), then the bench lines that show the polish family's time (cfg 3, cfg 2).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/polish_wave_r6.txt
echo "== $(date -u +%FT%TZ)" > $OUT
timeout 3000 python3 -m pytest tests -x -q -m gpu > gpurun_out/polish_wave_pytest.txt 2>&1
echo "parity: $(tail -1 gpurun_out/polish_wave_pytest.txt)" | tee -a $OUT
tail -15 gpurun_out/polish_wave_pytest.txt
short() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
sp=d['roofline']['serial_pass']['kernel_ms']
print('value %.1f ms_per_step %.1f serial %s' % (d['value'], d['ms_per_step'], {k:round(v,1) for k,v in sp.items()}))
"; }
B="--steps 10 --warmup 2 --no-cpu-baseline --no-shard8 --no-e2e"
echo "-- cfg3: $(timeout 600 python3 bench.py $B 2>/dev/null | short)" | tee -a $OUT
echo "-- cfg2: $(timeout 600 python3 bench.py --workload cfg2_ont_diploid_500x60x2kbp $B 2>/dev/null | short)" | tee -a $OUT
echo "-- cfg5: $(timeout 600 python3 bench.py --workload cfg5_hifi_diploid_2500x40x2kbp $B 2>/dev/null | short)" | tee -a $OUT
echo "-- cfg3: $(timeout 600 python3 bench.py $B 2>/dev/null | short)" | tee -a $OUT
