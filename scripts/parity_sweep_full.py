"""Diagnostic (GPU box): randomized full-path parity sweep (pair-HMM, polishing, variant filter, chain) against the
CPU oracle: consensus, re-threaded ops, labels, posteriors, scores.  Chunk ids (the RNG seeds) are random; every
fifth batch is a many-copy pile-up that takes clustering_recursive's split.
`python scripts/parity_sweep_full.py [n_batches] [seed]`"""
import sys
import numpy as np
import torch  # noqa: F401  (first: see bench.py)
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import api  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
bad = 0
for it in range(nb):
    config = str(rng.choice(["ont_diploid", "ont_diploid", "ont_noisy", "hifi_diploid", "ont_4copy"]))
    L = int(rng.integers(130, 1400))
    rph = int(rng.integers(3, 12))
    first = int(rng.integers(0, 1 << 40))
    kw = {}
    if it % 5 == 4:  # split branch: 6..10 copies present, copy_num 8..14 declared
        config, L, rph = "ont_4copy", int(rng.integers(400, 900)), int(rng.integers(6, 12))
        kw = dict(n_haps=int(rng.integers(6, 11)), copy_num=int(rng.integers(8, 15)), divergence=2e-2, min_variants=3)
    b, cfg, p = helpers.small_batch(config=config, n_chunks=3, tmpl_len=L, reads_per_hap=rph, first=first, **kw)
    dev = api.cluster_chunks(p, b, raise_on_chunk_failure=False)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    n, m = int(ora["cons_off"][-1]), int(ora["ops_out_off"][-1])
    ok = (np.array_equal(dev["cons_off"], ora["cons_off"]) and bytes(dev["cons"][:n]) == bytes(ora["cons"][:n])
          and np.array_equal(dev["ops_out_off"], ora["ops_out_off"]) and np.array_equal(dev["ops_out"][:m], ora["ops_out"][:m])
          and np.array_equal(dev["label"], ora["label"])
          and np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"]))
          and np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(ora["result"]["score"]))
          and np.array_equal(dev["result"]["polish_rounds"], ora["result"]["polish_rounds"]))
    print(it, config, "L", L, "reads/hap", rph, "rounds", dev["result"]["polish_rounds"].tolist(), "k",
          dev["result"]["cluster_num"].tolist(), "OK" if ok else "MISMATCH")
    bad += 0 if ok else 1
print("mismatching batches", bad, "of", nb)
