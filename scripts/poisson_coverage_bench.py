"""Diagnostic (GPU box): the headline workload with Poisson(60) reads per pile-up instead of exactly 60 -- what a real
60x data set looks like (a third of the pile-ups have more than 63 reads).  Prints kernel times of one pass."""
import sys
import numpy as np
import torch  # noqa: F401  (first: see bench.py)
sys.path.insert(0, "/root/repo")
from jtk_amd import api, batch as jb, synth

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(5)
base = dict(synth.CONFIGS["ont_diploid"])
piles = []
counts = []
for c in range(n_chunks):
    cfg = dict(base)
    n = max(8, int(rng.poisson(60)))
    cfg["reads_per_hap"] = max(2, n // 2)
    counts.append(2 * cfg["reads_per_hap"])
    piles.append(synth.make_pileup(c, cfg, synth.SEED0, 0))
b = jb.pack(piles)
p = jb.default_params(base["coverage"], base["band_frac"])
print("reads per pile-up: min %d median %d max %d, >63: %.0f%%" % (min(counts), int(np.median(counts)), max(counts),
                                                                   100.0 * np.mean(np.array(counts) > 63)))
with api.Session(p, b) as s:
    for _ in range(2):
        s.run()
        t = api.last_timing()
    out = s.fetch(raise_on_chunk_failure=False)
print({k: round(v, 1) for k, v in t["kernel_ms"].items()}, "total ms", round(t["total_ms"], 1), "chunks ok",
      int((out["result"]["status"] == 0).sum()), "chunks/s", round(n_chunks / (t["total_ms"] / 1e3), 1))
