#!/usr/bin/env python3
"""GPU box: the reference's trace! rows from the device (jtk_lc_session_trace) against the oracle's over a spread of pile-ups --
every candidate column's score and count, the pick order, every tried cluster count's score / expected gain / improved reads /
cluster sizes: intermediate values the results alone do not expose.  Prints one line per configuration and the totals.

    python3 scripts/trace_campaign.py [chunks per configuration, default 12]
"""
import os
import sys
import time

import torch  # noqa: F401  (first: see bench.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import api, batch as jb, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
CASES = [  # name, configuration, overrides, chunks, first chunk id
    ("cfg3 ont_diploid", "ont_diploid", {}, N, 7000),
    ("cfg5 hifi_diploid", "hifi_diploid", {}, N, 7100),
    ("ont_noisy 5/5/5 %", "ont_noisy", {}, max(2, N // 3), 7200),
    ("3 copies x 20 reads", "ont_4copy", dict(n_haps=3, copy_num=3, reads_per_hap=20), max(2, N // 3), 7300),
    ("cfg4 ont_4copy", "ont_4copy", {}, max(1, N // 6), 7400),
    ("diploid, 100 reads", "ont_diploid", dict(reads_per_hap=50), max(2, N // 4), 7500),
]


def main():
    O.build()
    tot_rows = tot_chunks = bad = 0
    t_dev = t_ora = 0.0
    for title, name, over, n_chunks, first in CASES:
        b, cfg = synth.make_batch(name, n_chunks, first_chunk_id=first, **over)
        p = jb.default_params(cfg["coverage"], cfg["band_frac"])
        po = helpers.oracle_params(p)
        rows_here = lk_rows = mism = 0
        with api.Session(p, b) as s:
            s.run()
            for c in range(n_chunks):
                t0 = time.perf_counter()
                dev = s.trace(c)
                t1 = time.perf_counter()
                _, ora = O.trace_chunk(po, b, c)
                t2 = time.perf_counter()
                t_dev += t1 - t0
                t_ora += t2 - t1
                rows_here += len(ora)
                lk_rows += sum(r.startswith("LK\t") for r in ora)
                if dev != ora:
                    mism += 1
                    for a, o in zip(dev, ora):
                        if a != o:
                            print("  MISMATCH %s chunk %d: device %r oracle %r" % (title, c, a, o))
                            break
                    else:
                        print("  MISMATCH %s chunk %d: %d device rows, %d oracle rows" % (title, c, len(dev), len(ora)))
        print("%-22s %3d chunks  %5d rows (%3d LK)  mismatching chunks: %d" % (title, n_chunks, rows_here, lk_rows, mism), flush=True)
        tot_rows += rows_here
        tot_chunks += n_chunks
        bad += mism
    print("total: %d chunks, %d rows, %d mismatching chunks; jtk_lc_session_trace %.1f s, oracle %.1f s" %
          (tot_chunks, tot_rows, bad, t_dev, t_ora))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
