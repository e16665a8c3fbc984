"""Diagnostic (GPU box): where does the device's modification table differ from the oracle's?  One 2 kbp ONT pile-up."""
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import api  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 4
b, cfg, p = helpers.small_batch(n_chunks=1, tmpl_len=L, reads_per_hap=nr)
reads = list(b.chunk_reads(0))
tab, lk = api.modification_table(p, b.template(0), [b.read(r) for r in reads], [b.read_ops(r) for r in reads],
                                 [b.strand[r] for r in reads])
po = helpers.oracle_params(p)
radius = int(np.ceil(len(b.template(0)) * cfg["band_frac"])) // 2
for k, r in enumerate(reads):
    h = po.forward if b.strand[r] else po.reverse
    ot, olk = O.modification_table(h, b.template(0), b.read(r), b.read_ops(r), radius)
    ot = ot - olk
    d = tab[k].view(np.uint64) != ot.view(np.uint64)
    Lt = len(b.template(0))
    d2 = d.reshape(Lt + 1, 14)
    print(f"read {k}: len {len(b.read(r))} lk dev {lk[k]!r} ora {olk!r}; differing entries {int(d.sum())} of {d.size}")
    if d.any():
        pos = np.nonzero(d2.any(axis=1))[0]
        print("  positions:", pos[:12], "...", pos[-6:], " count", len(pos))
        print("  by table row:", d2.sum(axis=0).tolist())
        p0 = pos[0]
        print("  first pos", p0, "dev", tab[k].reshape(Lt + 1, 14)[p0], "\n            ora", ot.reshape(Lt + 1, 14)[p0])
        runs = np.split(pos, np.nonzero(np.diff(pos) > 1)[0] + 1)
        print("  runs:", [(int(r_[0]), int(r_[-1])) for r_ in runs[:20]])
