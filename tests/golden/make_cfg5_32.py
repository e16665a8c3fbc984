"""Generates tests/golden/cfg5_32.npz: the CPU oracle's results on 32 pile-ups of BASELINE cfg 5 at its defining shape (HiFi
error model 0.1 %, 2 haplotypes x 20 reads x 2 kbp = 40 reads per pile-up, band fraction 0.01 -> radius 10: the path that runs on
phmm_pair_kernel), chunk ids 5300 .. 5331, at least one variant column each.  Inputs come from jtk_synth_pileup (seeds 20260101 +
chunk id), so only the EXPECTED outputs -- incl. the re-threaded ops, which the pair kernel's band follows -- and a checksum of
the inputs are stored.  Self-consistency vectors (oracle == device) like cfg3_64.npz: the device path is checked against them
without the oracle in the loop (tests/test_gpu_defining_shapes.py).  Re-run to regenerate (~1 min on 8 cores)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import batch as jb, build, synth  # noqa: E402
from make_cfg3_64 import inputs_digest  # noqa: E402

FIRST, COUNT = 5300, 32


def make_inputs(first=FIRST, count=COUNT):
    b, cfg = synth.make_batch("hifi_diploid", count, first_chunk_id=first, min_variants=1)
    return b, cfg, jb.default_params(cfg["coverage"], cfg["band_frac"])


def main():
    build.build()
    b, cfg, p = make_inputs()
    out = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert out["rc"] == 0
    nc, no = int(out["cons_off"][-1]), int(out["ops_out_off"][-1])
    np.savez_compressed(os.path.join(HERE, "cfg5_32.npz"), inputs_sha256=np.array([inputs_digest(b)]), label=out["label"],
                        log_post=out["log_post"], result=out["result"], cons=out["cons"][:nc], cons_off=out["cons_off"],
                        ops_out=out["ops_out"][:no], ops_out_off=out["ops_out_off"],
                        oracle_sha256=np.array([helpers.oracle_sources_sha()]))
    print("wrote cfg5_32.npz: k =", np.bincount(out["result"]["cluster_num"]).tolist(), "n_variants",
          out["result"]["n_variants"].tolist(), "rounds", out["result"]["polish_rounds"].tolist())


if __name__ == "__main__":
    main()
