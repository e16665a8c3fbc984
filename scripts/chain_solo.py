"""Diagnostic (GPU box; run directly under rocprofv3): every listed chunk of cfg 3 through the stage ALONE, one call per chunk, in
the order given -- the chain kernel's dispatches then appear in that order in a kernel trace / counter collection.
    python3 scripts/chain_solo.py --chunks 600 --ids 591,293 [--refit]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jtk_amd import api, batch as jb, ffi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=600)
    ap.add_argument("--ids", required=True)
    ap.add_argument("--refit", action="store_true")
    args = ap.parse_args()
    b, cfg = synth.make_batch("ont_diploid", args.chunks)
    p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
    if args.refit:
        n = b.chunks["n_reads"].astype(np.int64)
        cov = int(np.sort(n)[len(n) // 2])
        by_id = np.argsort(b.chunks["chunk_id"], kind="stable")
        train = [int(c) for c in by_id if max(cov, 2) - 2 <= n[c] < cov + 2][:5]
        f, r = api.fit_model(p, b.subset(train), rounds=10)
        q = ffi.Params.from_buffer_copy(bytes(p))
        q.forward, q.reverse = f, r
        q.gains = api.estimate_gains(f, r)
        p = q
    for c in [int(x) for x in args.ids.split(",")]:
        o = api.cluster_chunks(p, b.subset([c]))
        print("SOLO chunk %d n_variants %d mcmc_ms %.2f" % (c, int(o["result"]["n_variants"][0]), api.last_timing()["kernel_ms"]["mcmc"]))


if __name__ == "__main__":
    main()
