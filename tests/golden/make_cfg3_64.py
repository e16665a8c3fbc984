"""Generates tests/golden/cfg3_64.npz: the CPU oracle's results on the first 64 pile-ups of the headline workload (cfg 3:
2 kbp x 60 ONT reads, diploid; inputs come from jtk_synth_pileup, seeds 20260101 + chunk id, so only the EXPECTED outputs and
a checksum of the inputs are stored).  Self-consistency vectors (oracle == device), like small_pileups.npz: the device path is
checked against them without the oracle in the loop (tests/test_gpu_defining_shapes.py).  Re-run to regenerate (~2 min)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import batch as jb, build, synth  # noqa: E402


def inputs_digest(b):
    h = hashlib.sha256()
    for a in (b.chunks, b.tmpl_bases, b.read_bases, b.read_off, b.ops, b.ops_off, b.strand):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    build.build()
    b, cfg = synth.make_batch("ont_diploid", 64)
    p = jb.default_params(cfg["coverage"], cfg["band_frac"])
    out = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert out["rc"] == 0
    nc = int(out["cons_off"][-1])
    np.savez_compressed(os.path.join(HERE, "cfg3_64.npz"), inputs_sha256=np.array([inputs_digest(b)]), label=out["label"],
                        log_post=out["log_post"], result=out["result"], cons=out["cons"][:nc], cons_off=out["cons_off"])
    print("wrote cfg3_64.npz: k =", np.bincount(out["result"]["cluster_num"]).tolist(), "rounds mean",
          float(out["result"]["polish_rounds"].mean()))


if __name__ == "__main__":
    main()
