#!/bin/bash
# Diagnostic (GPU box): rebuild with per-chunk chain counters, run one bench pass, keep the slowest chunks.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
export JTK_LC_LIB=$(python3 -c "import jtk_amd.build as b; print(b.build_experiment('probe1', '''-DJTK_MCMC_STATS'''))") || exit 1
python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/stats_raw.txt 2>&1
python3 - <<'PY'
import re, numpy as np
rows=[]
for line in open("gpurun_out/stats_raw.txt"):
    for m in re.finditer(r"K2STAT chunk (\d+) n (\d+) D (\d+) cyc (\d+) walk \d+ win (\d+) event \d+ rebuild (\d+) steps (\d+) windows (\d+) events (\d+) accepts (\d+) changed (\d+)", line):
        rows.append([int(x) for x in m.groups()])
a=np.array(rows,dtype=float)   # chunk n D cyc inline_size_moves cyc_general steps windows events accepts changed
print("chunks with stats:",len(a))
for D in sorted(set(a[:,2])):
    s=a[a[:,2]==D]
    X=np.stack([np.ones(len(s)),s[:,8]],1); y=s[:,3]
    coef,*_=np.linalg.lstsq(X,y,rcond=None)
    print("D=%d n=%d: cycles ~ %.4g (%.1f/step) + %.0f/event; mean events %.0f, inline size-only moves %.0f, max cyc %.3g"%(D,len(s),coef[0],coef[0]/s[:,6].mean(),coef[1],s[:,8].mean(),s[:,4].mean(),y.max()))
w=[int(m.group(2)) for line in open('gpurun_out/stats_raw.txt') for m in re.finditer(r'K2WAIT chunk (\d+) waits (\d+)', line)]
print('record-wait polls per chunk: mean %.0f max %d'%(np.mean(w) if w else 0, max(w) if w else 0))
pr=[[int(x) for x in m.groups()] for line in open('gpurun_out/stats_raw.txt') for m in re.finditer(r'K2PROD wr (\d+) sleeps (\d+) cyc_gen (\d+) cyc_parse (\d+) cyc_jump (\d+)', line)]
if pr:
    q=np.array(pr,dtype=float); sb=q[:,0]/2048
    print('producer per superblock (2048 draws): gen %.0f parse %.0f jump %.0f cycles; sleeps/chunk %.0f'%((q[:,2]/sb).mean(),(q[:,3]/sb).mean(),(q[:,4]/sb).mean(),q[:,1].mean()))
i=np.argsort(-a[:,3])[:6]
for r in a[i]: print("slow chunk %d D %d cyc %.4g events %d (cycles in general events %.3g) inline %d windows %d"%(r[0],r[2],r[3],r[8],r[5],r[4],r[7]))
PY
unset JTK_LC_LIB   # the product library was never touched
