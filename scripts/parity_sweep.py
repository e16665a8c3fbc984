"""Diagnostic (GPU box): randomized feature-level parity sweep of the chain kernels against the CPU oracle --
labels, posteriors and scores bit for bit.  `python scripts/parity_sweep.py [n_seeds] [first_seed]`"""
import sys
import numpy as np
import torch  # noqa: F401  (first: see bench.py)
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import helpers  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from jtk_amd import batch as jb  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
first_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = total = 0
for seed in range(first_seed, first_seed + n_seeds):
    rng = np.random.default_rng(seed)
    specs = []
    for _ in range(14):
        cn = int(rng.choice([2, 2, 2, 2, 3, 4]))
        n = int(rng.integers(4, 128)) if cn == 2 else int(rng.integers(6, 90))
        d = int(rng.integers(1, 9)) if cn == 2 else int(rng.integers(1, 3 * cn + 1))
        specs.append((n, d, int(rng.integers(1, cn + 1)), cn))
    p = jb.default_params(haploid_coverage=float(rng.choice([8.0, 15.0, 30.0])))
    dev, ora, truth = T.run_features_both(p, specs, seed=seed)
    off = 0
    for i, (n, d, kt, cn) in enumerate(specs):
        ok = (np.array_equal(dev["label"][off:off + n], ora["label"][off:off + n])
              and np.array_equal(helpers.bits(dev["log_post"][off:off + n]), helpers.bits(ora["log_post"][off:off + n]))
              and helpers.bits(dev["result"]["score"][i:i + 1])[0] == helpers.bits(ora["result"]["score"][i:i + 1])[0]
              and dev["result"]["cluster_num"][i] == ora["result"]["cluster_num"][i])
        total += 1
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "chunk", i, specs[i], "score", dev["result"]["score"][i], ora["result"]["score"][i])
        off += n
print("mismatches", bad, "of", total)
