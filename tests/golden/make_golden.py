"""Generates tests/golden/small_pileups.npz: five small seeded pile-ups (inputs in the flat C-ABI layout) with the
outputs of the CPU oracle for them.  The reference (Rust + un-vendored kiley) cannot be built or imported here, so
these are SELF-CONSISTENCY vectors (oracle == oracle over time, GPU == oracle), not reference-parity vectors
(SURVEY.md 8c).  Inputs come from jtk_synth_pileup with the seeds below; re-run to regenerate."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import batch as jb, build, synth  # noqa: E402

build.build()
cfgs = [("ont_diploid", 300, 12, 2), ("ont_diploid", 360, 12, 2), ("ont_4copy", 320, 10, 4), ("hifi_diploid", 400, 12, 2),
        ("ont_noisy", 260, 12, 2)]
piles = []
for i, (name, L, rph, cn) in enumerate(cfgs):
    cfg = dict(synth.CONFIGS[name])
    cfg.update(tmpl_len=L, reads_per_hap=rph)
    piles.append(synth.make_pileup(100 + i, cfg, min_variants=2))
b = jb.pack(piles)
p = jb.default_params(haploid_coverage=12.0, band_frac=0.03)
full = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
pol = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=True)
assert full["rc"] == 0 and pol["rc"] == 0
nc, nm = int(full["cons_off"][-1]), int(full["ops_out_off"][-1])
np.savez_compressed(
    os.path.join(HERE, "small_pileups.npz"),
    chunks=b.chunks, tmpl_bases=b.tmpl_bases, read_bases=b.read_bases, read_off=b.read_off, ops=b.ops,
    ops_off=b.ops_off, strand=b.strand, haploid_coverage=np.array([12.0]), band_frac=np.array([0.03]),
    label=full["label"], log_post=full["log_post"], result=full["result"], cons=full["cons"][:nc],
    cons_off=full["cons_off"], ops_out=full["ops_out"][:nm], ops_out_off=full["ops_out_off"],
    polished_label=pol["label"], polished_log_post=pol["log_post"], polished_result=pol["result"])
print("wrote", os.path.join(HERE, "small_pileups.npz"), "k =", full["result"]["cluster_num"].tolist(),
      "rounds =", full["result"]["polish_rounds"].tolist(), "D =", full["result"]["n_variants"].tolist())
