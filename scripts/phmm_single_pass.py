"""Diagnostic: time of ONE pair-HMM pass over the bench batch (skip_polish), for kernel experiments."""
import sys
import torch  # noqa: F401  (first: see bench.py)
sys.path.insert(0, "/root/repo")
from jtk_amd import api, batch as jb, synth

cfg = dict(synth.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "ont_diploid"])  # e.g. hifi_diploid: the pair kernel
b, cfg = synth.make_batch(cfg, int(sys.argv[1]) if len(sys.argv) > 1 else 500)
p = jb.default_params(cfg["coverage"], cfg["band_frac"])
s = api.Session(p, b)
for _ in range(2):
    try:
        s.run(skip_polish=True)
    except Exception as e:  # garbage tables may fail chunks; timing is still recorded
        print("run:", e)
    t = api.last_timing()
    print({k: round(v, 2) for k, v in t["kernel_ms"].items()})
