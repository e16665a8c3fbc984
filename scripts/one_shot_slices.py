"""Diagnostic (GPU box): wall time of the one-shot jtk_lc_cluster_chunks on a 2500-chunk batch, unsliced and with
the automatic slicing (up to 3 slices on their own streams / host threads).  `python scripts/one_shot_slices.py [n_chunks]`"""
import os
import sys
import time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, "/root/repo")
from jtk_amd import api, batch as jb  # noqa: E402
from bench import make_batch_parallel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
b, cfg = make_batch_parallel("ont_diploid", n, 0)
p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
ref = None
for slices in ("1", "1", "1", "2", "2", "2", "3", "3", "3", "", ""):   # repeated: the block pool settles per shape
    if slices:
        os.environ["JTK_LC_SLICES"] = slices
    else:
        os.environ.pop("JTK_LC_SLICES", None)
    t0 = time.perf_counter()
    out = api.cluster_chunks(p, b)
    dt = time.perf_counter() - t0
    if ref is None:
        ref = out
    same = all(np.array_equal(out[k], ref[k]) for k in ("label", "log_post", "cons", "ops_out", "ops_out_off"))
    print("JTK_LC_SLICES=%-5s %d chunks in %.2f s (upload, run, fetch) -> %.0f chunks/s, identical %s"
          % (slices or "auto", n, dt, n / dt, same))
