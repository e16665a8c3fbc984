"""Diagnostic (GPU box): the one-shot stage call (host buffers in, host buffers out) on the headline data set with
different numbers of slices (JTK_LC_SLICES), second call each (workspaces pooled).  `python scripts/one_shot_slices.py`"""
import os
import sys
import time
os.environ.setdefault("JTK_LC_POOL_GB", "160")
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, "/root/repo")
from jtk_amd import api, batch as jb, synth  # noqa: E402
from bench import make_batch_parallel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
b, cfg = make_batch_parallel("ont_diploid", np.arange(n))
p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
ref = None
for slices in (3, 4, 5, 6, 8):
    os.environ["JTK_LC_SLICES"] = str(slices)
    best = 1e9
    for it in range(3):
        t0 = time.perf_counter()
        out = api.cluster_chunks(p, b)
        best = min(best, time.perf_counter() - t0)
    if ref is None:
        ref = out["label"].copy()
    print("slices %d: %.3f s = %.0f chunks/s, labels equal %s, timing %s" % (
        slices, best, n / best, bool(np.array_equal(ref, out["label"])),
        {k: round(v) for k, v in api.last_timing()["kernel_ms"].items()}))
