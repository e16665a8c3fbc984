#!/bin/bash
# GPU box, round 6: the reference's trace! rows from the device (jtk_lc_session_trace) -- the tests that compare them with the
# oracle's, the JSON drop-in's --trace, a sanity pass of the goldens on the library that carries the recording instantiations,
# and one 4-copy chunk's rows with the time the call takes (-> profiles/r06_trace_rows.txt).
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/trace_rows_r6.txt
echo "== $(date -u +%FT%TZ) library $(python3 -c "import bench; print(bench.lib_sha16())" 2>/dev/null)" > $OUT
timeout 900 python3 -m pytest tests/test_trace_rows.py -x -q -m gpu > gpurun_out/trace_rows_r6_pytest.txt 2>&1
echo "trace tests: $(tail -1 gpurun_out/trace_rows_r6_pytest.txt)" | tee -a $OUT
echo "json drop-in (--trace): $(timeout 900 python3 -m pytest tests/test_dataset_json.py -x -q -m gpu 2>&1 | tail -1)" | tee -a $OUT
echo "goldens: $(timeout 1200 python3 -m pytest tests/test_golden.py tests/test_gpu_defining_shapes.py -x -q -m gpu -k 'golden' 2>&1 | tail -1)" | tee -a $OUT
timeout 600 python3 - >> $OUT 2>&1 <<'PY'
import time
import torch  # noqa: F401
from jtk_amd import api, batch as jb, synth
for name, c in (("ont_4copy", 0), ("ont_diploid", 0)):
    b, cfg = synth.make_batch(name, 2)
    p = jb.default_params(cfg["coverage"], cfg["band_frac"])
    with api.Session(p, b) as s:
        s.run()
        t0 = time.perf_counter()
        rows = s.trace(c)
        dt = time.perf_counter() - t0
    print("-- %s chunk %d: %d rows, jtk_lc_session_trace took %.0f ms" % (name, c, len(rows), dt * 1e3))
    print("\n".join(rows))
PY
cat $OUT
