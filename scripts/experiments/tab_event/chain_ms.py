"""Diagnostic (GPU box): the chain kernels' device time on N full-size 4-copy pile-ups (cfg 4: 160 reads, 2 kbp), product library.
`python3 scripts/experiments/tab_event/chain_ms.py [n_chunks]` -- round 4 before the LDS size table: 10,337 ms for 8 chunks
(profiles/r04_chain_stats_tab.txt, first line)."""
import os
import sys
import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from jtk_amd import api, batch as jb, synth  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b, cfg = synth.make_batch("ont_4copy", n)
p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
with api.Session(p, b) as s:
    s.run()
    t = api.last_timing()
    r = s.fetch_results()
print("MCMCMS", t["kernel_ms"]["mcmc"], "k", r["result"]["cluster_num"].tolist(), "D", r["result"]["n_variants"].tolist())
