#!/bin/bash
# Diagnostic (GPU box): kernel trace of three headline steps -> gpurun_out/trace_r6.csv (one row per dispatch) + its timeline summary
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_r6 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shard8 --no-e2e "$@" > /dev/null 2>&1 )
python3 scripts/trace_timeline.py gpurun_out/trace_r6 2>&1 | tee gpurun_out/trace_r6_summary.txt
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/trace_r6/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            n = n.replace("(anonymous namespace)::", "").split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""), n, r.get("Grid_Size_X", r.get("Grid_Size", ""))))
rows.sort()
t0 = rows[0][0] if rows else 0
with open("gpurun_out/trace_r6.csv", "w") as out:
    out.write("start_us,end_us,queue,kernel,grid_x\n")
    for a, b, q, n, g in rows:
        out.write("%d,%d,%s,%s,%s\n" % ((a - t0) // 1000, (b - t0) // 1000, q, n, g))
print("rows", len(rows))
PY
rm -rf gpurun_out/trace_r6
