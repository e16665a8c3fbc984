"""Generates tests/golden/cfg3_refit_32.npz: the stage AS JTK ENTERS IT on 32 pile-ups of the headline workload -- the model is
refitted on the stage's own training pile-ups (update_models_on_both_strands, model_tune.rs:96-156: the first five by id whose
coverage is within 2 of the median, TRAIN_ROUND = 10) and the gains are calibrated on the refitted model (estimate_gain_default,
likelihood_gains.rs:186-192, mod.rs:58-60), all by the CPU oracle.  The refitted model lets weak variant columns through the
filter; on such chunks the Metropolis chain accepts 10^5 .. 10^6 moves -- the regime the table-driven walk is most likely to get
wrong -- so the sample is the first 24 chunks plus eight eventful ones (bench.REFIT_EVENTFUL; profiles/r05_chain_pieces_before.txt).
Inputs come from jtk_synth_pileup (seeds 20260101 + chunk id): only the refitted parameters, the expected outputs and a checksum
of the inputs are stored.  Self-consistency vectors (oracle == device) like cfg3_64.npz: checked on the device without the oracle
in the loop (tests/test_gpu_defining_shapes.py).  Re-run to regenerate (~10 min on 8 CPUs)."""
import ctypes as C
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import batch as jb, build, synth  # noqa: E402
from make_cfg3_64 import inputs_digest  # noqa: E402

EVENTFUL = (196, 206, 269, 332, 367, 464, 478, 498)   # == bench.REFIT_EVENTFUL
CHUNK_IDS = tuple(range(24)) + EVENTFUL


def make_inputs():
    cfg = dict(synth.CONFIGS["ont_diploid"])
    b = jb.pack([synth.make_pileup(int(c), cfg) for c in CHUNK_IDS])
    return b, cfg


def training_chunks(b):
    """update_models_on_both_strands' choice (model_tune.rs:99-118) on this batch: every pile-up has 60 reads, so it is the first
    five by id -- the same five the full 2,500-chunk data set trains on."""
    n = b.chunks["n_reads"].astype(np.int64)
    cov = int(np.sort(n)[len(n) // 2])
    by_id = np.argsort(b.chunks["chunk_id"], kind="stable")
    return [int(c) for c in by_id if max(cov, 2) - 2 <= n[c] < cov + 2][:5]


def main():
    build.build()
    b, cfg = make_inputs()
    p = jb.default_params(cfg["coverage"], cfg["band_frac"])
    po = helpers.oracle_params(p)
    t0 = time.time()
    rc, f, r = O.fit_model(po, b.subset(training_chunks(b)), rounds=10)
    assert rc == 0
    po.forward, po.reverse = f, r
    O.lib().jo_estimate_gain_default(C.byref(po.forward), C.byref(po.reverse), C.byref(po.gains))
    print("refit + gains: %.0f s" % (time.time() - t0))
    out = O.cluster_chunks(po, b, skip_polish=False)
    assert out["rc"] == 0
    nc = int(out["cons_off"][-1])
    np.savez_compressed(os.path.join(HERE, "cfg3_refit_32.npz"), inputs_sha256=np.array([inputs_digest(b)]),
                        chunk_ids=np.array(CHUNK_IDS), params=np.frombuffer(bytes(po), dtype=np.uint8),
                        label=out["label"], log_post=out["log_post"], result=out["result"], cons=out["cons"][:nc],
                        cons_off=out["cons_off"])
    print("wrote cfg3_refit_32.npz: k =", np.bincount(out["result"]["cluster_num"]).tolist(), "variants",
          out["result"]["n_variants"].tolist(), "total %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
