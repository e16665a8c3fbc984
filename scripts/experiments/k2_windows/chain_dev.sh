#!/bin/bash
# Diagnostic (GPU box): chain parity (feature-level sweep + the chain tests) and the chain kernel's time on one 625-chunk
# slice of the headline workload, with the per-chunk counters of a -DJTK_MCMC_STATS build beside it.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
if [ "${1:-6}" != "0" ]; then
python3 scripts/parity_sweep.py ${1:-6} 100 2>&1 | tail -5
timeout 900 python3 -m pytest -x -q tests/test_gpu_parity.py -k "chain_variants or size_only or cluster_features or edge_cases" tests/test_gpu_shapes.py::test_random_chain_sweep_matches_oracle 2>&1 | tail -3
fi
cat > /tmp/chain_time.py <<'PY'
import sys, time
import torch  # noqa
sys.path.insert(0, ".")
from jtk_amd import api, batch as jb, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 625
b, cfg = synth.make_batch("ont_diploid", n)
p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
with api.Session(p, b) as s:
    for it in range(2):
        s.run()
        t = api.last_timing()
        print("CHAINMS n=%d run %d" % (n, it), {k: round(v, 1) for k, v in t["kernel_ms"].items()})
PY
python3 /tmp/chain_time.py 625 2>&1 | grep CHAINMS
python3 /tmp/chain_time.py 125 2>&1 | grep CHAINMS
if [ "${2:-stats}" = "stats" ]; then
export JTK_LC_LIB=$(python3 -c "import jtk_amd.build as b; print(b.build_experiment('probe1', '''-DJTK_MCMC_STATS'''))") || exit 1
python3 /tmp/chain_time.py 300 > gpurun_out/k2stats_raw.txt 2>&1
python3 - <<'PY'
import re, numpy as np
rows=[]
for line in open("gpurun_out/k2stats_raw.txt", errors="replace"):
    for m in re.finditer(r"K2STAT chunk (\d+) n (\d+) D (\d+) cyc (\d+) walk (\d+) win (\d+) event (\d+) rebuild (\d+) steps (\d+) windows (\d+) events (\d+) accepts (\d+) changed (\d+) iters (\d+) cyc_top (\d+) cyc_fast (\d+) gathers (\d+) settled (\d+)", line):
        rows.append([int(x) for x in m.groups()])
a=np.array(rows,dtype=float)
if len(a):
    # columns: chunk n D cyc [1] nullcnt cyc_null cyc_general steps windows events accepts changed
    print("chunks with stats:", len(a), " (two runs each)")
    print("cycles per chunk: median %.3g mean %.3g max %.3g" % (np.median(a[:,3]), a[:,3].mean(), a[:,3].max()))
    quiet=a[a[:,10] < 2000]
    if len(quiet): print("quiet chunks (<2000 events): %d, cycles/proposal %.1f" % (len(quiet), (quiet[:,3]/quiet[:,8]).mean()))
    print("window builds: cycles per window %.0f" % (a[:,4].sum()/max(1,a[:,9].sum())))
    print("null moves: mean %.0f, cycles each %.0f" % (a[:,5].mean(), a[:,6].sum()/max(1,a[:,5].sum())))
    gen=a[:,10]-a[:,5]
    print("general events: mean %.0f, cycles each %.0f" % (gen.mean(), a[:,7].sum()/max(1,gen.sum())))
    print("loop iterations: mean %.0f; cycles top->hit each %.0f; fast size-only accepts (top->continue) each %.0f; regathers mean %.0f; hits settled by a scalar test mean %.0f" % (a[:,13].mean(), a[:,14].sum()/max(1,a[:,13].sum()), a[:,15].sum()/max(1,(a[:,5]).sum()), a[:,16].mean(), a[:,17].mean()))
    i=np.argsort(-a[:,3])[:5]
    for r in a[i]: print("slow chunk %d D %d cyc %.4g null %d general %d accepts %d changed %d windows %d" % (r[0],r[2],r[3],r[5],r[10]-r[5],r[11],r[12],r[9]))
pr=[[int(x) for x in m.groups()] for line in open('gpurun_out/k2stats_raw.txt', errors="replace") for m in re.finditer(r'K2PROD wr (\d+) sleeps (\d+) cyc_gen (\d+) cyc_parse (\d+) cyc_jump (\d+)', line)]
if pr:
    q=np.array(pr,dtype=float); sb=q[:,0]/1024
    print('producer per 1024 draws: gen %.0f parse %.0f jump %.0f cycles; sleeps/chunk %.0f'%((q[:,2]/sb).mean(),(q[:,3]/sb).mean(),(q[:,4]/sb).mean(),q[:,1].mean()))
w=[int(m.group(2)) for line in open('gpurun_out/k2stats_raw.txt', errors="replace") for m in re.finditer(r'K2WAIT chunk (\d+) waits (\d+)', line)]
if w: print('record-wait polls per chunk: mean %.0f max %d'%(np.mean(w), max(w)))
PY
unset JTK_LC_LIB
fi
