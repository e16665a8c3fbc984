"""Diagnostic (GPU box): where a one-shot call spends its wall time: session create (encode, hipMalloc, upload), run,
fetch, destroy.  `python scripts/one_shot_phases.py [n_chunks]`"""
import sys
import time
import torch  # noqa: F401
sys.path.insert(0, "/root/repo")
from jtk_amd import api, batch as jb  # noqa: E402
from bench import make_batch_parallel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
b, cfg = make_batch_parallel("ont_diploid", n, 0)
p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
for it in range(2):
    t0 = time.perf_counter()
    s = api.Session(p, b)
    t1 = time.perf_counter()
    s.run()
    t2 = time.perf_counter()
    out = s.fetch()
    t3 = time.perf_counter()
    s.close()
    t4 = time.perf_counter()
    print("pass %d: create %.2f s, run %.2f s, fetch %.2f s, destroy %.2f s; h2d %.0f ms"
          % (it, t1 - t0, t2 - t1, t3 - t2, t4 - t3, api.last_timing()["h2d_ms"]))
