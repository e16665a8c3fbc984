"""Generates tests/golden/cfg4_16.npz: the CPU oracle's results on 16 pile-ups of BASELINE cfg 4 at its defining shape (4 copies
x 40 reads x 2 kbp = 160 reads per pile-up, candidate k = 2, 3, 4: every chain of the K-way table kernel), chunk ids 4200 ..
4215 with at least two variant columns.  Inputs come from jtk_synth_pileup (seeds 20260101 + chunk id), so only the EXPECTED
outputs and a checksum of the inputs are stored.  Self-consistency vectors (oracle == device) like cfg3_64.npz: the device path
is checked against them without the oracle in the loop (tests/test_gpu_defining_shapes.py).  Re-run to regenerate (~4 min on
8 cores)."""
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

FIRST, COUNT = 4200, 16


def make_inputs():
    b, cfg = synth.make_batch("ont_4copy", COUNT, first_chunk_id=FIRST, min_variants=2)
    return b, cfg, jb.default_params(cfg["coverage"], cfg["band_frac"])


def main():
    build.build()
    b, cfg, p = make_inputs()
    out = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert out["rc"] == 0
    nc = int(out["cons_off"][-1])
    np.savez_compressed(os.path.join(HERE, "cfg4_16.npz"), inputs_sha256=np.array([inputs_digest(b)]), label=out["label"],
                        log_post=out["log_post"], result=out["result"], cons=out["cons"][:nc], cons_off=out["cons_off"])
    print("wrote cfg4_16.npz: k =", np.bincount(out["result"]["cluster_num"]).tolist(), "n_variants",
          out["result"]["n_variants"].tolist())


if __name__ == "__main__":
    main()
