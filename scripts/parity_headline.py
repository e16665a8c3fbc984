"""Diagnostic (GPU box): bit-parity of the full path with the CPU oracle AT THE HEADLINE SIZE (2 kbp, 30 reads per
haplotype) on chunks with random ids, i.e. RNG streams the committed tests never saw.
`python scripts/parity_headline.py [n_chunks] [seed] [config]`"""
import sys
import time
import numpy as np
import torch  # noqa: F401  (first: see bench.py)
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import api, batch as jb, synth  # noqa: E402

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
config = sys.argv[3] if len(sys.argv) > 3 else "ont_diploid"
first = int(np.random.default_rng(seed).integers(0, 1 << 40))
b, cfg = synth.make_batch(config, n_chunks, first_chunk_id=first)
p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
t0 = time.time()
dev = api.cluster_chunks(p, b, raise_on_chunk_failure=False)
t1 = time.time()
ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
t2 = time.time()
n, m = int(ora["cons_off"][-1]), int(ora["ops_out_off"][-1])
checks = {
    "consensus": np.array_equal(dev["cons_off"], ora["cons_off"]) and bytes(dev["cons"][:n]) == bytes(ora["cons"][:n]),
    "ops": np.array_equal(dev["ops_out_off"], ora["ops_out_off"]) and np.array_equal(dev["ops_out"][:m], ora["ops_out"][:m]),
    "labels": np.array_equal(dev["label"], ora["label"]),
    "cluster_num": np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"]),
    "posterior bits": np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"])),
    "score bits": np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(ora["result"]["score"])),
    "polish rounds": np.array_equal(dev["result"]["polish_rounds"], ora["result"]["polish_rounds"]),
}
bad_chunks = [c for c in range(n_chunks)
              if not np.array_equal(dev["label"][list(b.chunk_reads(c))], ora["label"][list(b.chunk_reads(c))])]
print(config, "chunks", n_chunks, "first id", first, "device %.1fs oracle %.1fs" % (t1 - t0, t2 - t1))
print("k histogram", np.bincount(ora["result"]["cluster_num"]).tolist(), "status", int(np.abs(dev["result"]["status"]).sum()))
print(checks, "label-mismatching chunks", bad_chunks[:10])
print("PARITY", "OK" if all(checks.values()) else "MISMATCH")
