"""Generates tests/golden/trace_rows.json: the reference's trace! rows (TOTAL / CAND / PICK / DUMP / RANGE / LK / COUNTS;
pseudo_mcmc.rs:122-127,236,250-262,467-472,539) of a few pile-ups as the CPU oracle writes them (oracle/pseudo_mcmc.c) -- diploid
ONT with and without a variant column, HiFi, a 4-copy pile-up whose model selection tries k = 2, 3, 4.  Inputs come from
jtk_synth_pileup, so only the rows and a checksum of the inputs are stored.  The device's jtk_lc_session_trace is compared with the
file without the oracle in the loop (tests/test_trace_rows.py), the oracle with it on the CPU.  Re-run to regenerate (~30 s)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import batch as jb, build, synth  # noqa: E402
from make_cfg3_64 import inputs_digest  # noqa: E402

CASES = [("ont_diploid", 3, [0, 1, 2]), ("hifi_diploid", 1, [0]), ("ont_4copy", 1, [0])]   # config, chunks made, chunks traced


def make_inputs(name, n_chunks):
    b, cfg = synth.make_batch(name, n_chunks)
    return b, jb.default_params(cfg["coverage"], cfg["band_frac"])


def main():
    build.build()
    out = []
    for name, n_chunks, which in CASES:
        b, p = make_inputs(name, n_chunks)
        for c in which:
            res, rows = O.trace_chunk(helpers.oracle_params(p), b, c)
            assert res["rc"] == 0
            out.append(dict(config=name, n_chunks=n_chunks, chunk=c, inputs_sha256=inputs_digest(b), rows=rows))
            print(name, c, len(rows), "rows")
    with open(os.path.join(HERE, "trace_rows.json"), "w") as fh:
        json.dump(dict(oracle_sha256=helpers.oracle_sources_sha(), cases=out), fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
