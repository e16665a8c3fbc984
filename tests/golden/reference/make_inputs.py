"""Writes tests/golden/reference/inputs.json: the inputs rust/dump_golden.rs (run inside a ban-m/jtk checkout, with the real
kiley / rand / nalgebra crates) turns into tests/golden/reference/reference_golden.json, which tests/test_reference_golden.py
checks the CPU oracle and the device path against.  Everything here is DATA produced by this repository's own generators
(seeded): the RNG call parameters this path uses, the feature matrices of
tests/test_gpu_parity.py::test_cluster_features_matches_oracle, the five pile-ups of tests/golden/small_pileups.npz, the
default model, and two graph Laplacians.  Re-run to regenerate:  python tests/golden/reference/make_inputs.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(TESTS))
sys.path.insert(0, TESTS)
import helpers  # noqa: E402

def fbits(xs):
    """f64 values as their bit patterns (exact through any JSON parser)"""
    return [int(v) for v in np.ascontiguousarray(xs, dtype=np.float64).view(np.uint64).ravel()]


OPS = "=XID"  # include/jtk_lc.h: 0 Match, 1 Mismatch, 2 Ins, 3 Del
FEATURE_SPECS = [(24, 3, 2, 2), (24, 6, 2, 2), (30, 1, 2, 2), (36, 5, 3, 3), (40, 6, 3, 4), (20, 4, 1, 2), (6, 3, 2, 2),
                 (2, 2, 2, 2), (25, 0, 2, 2), (20, 3, 2, 1), (60, 4, 2, 2), (160, 9, 4, 4)]
FEATURE_SEED = 1
FEATURE_COVERAGE = 12.0


def hmm_dict():
    h = {k: 0.97 for k in ("mat_mat", "ins_mat", "del_mat")}
    h.update({k: 0.01 for k in ("mat_ins", "mat_del", "ins_ins", "ins_del", "del_ins", "del_del")})
    h["mat_emit"] = [0.97 if r == q else 0.01 for r in range(4) for q in range(4)]  # definitions/src/lib.rs:128-147
    h["ins_emit"] = [0.25] * 20
    return h


def feature_problems():
    rng = np.random.default_rng(FEATURE_SEED)
    out = []
    for i, (n, dim, k_true, copy_num) in enumerate(FEATURE_SPECS):
        x, vt, _ = helpers.random_feature_problem(rng, n, dim, k_true)
        out.append(dict(name=f"n{n}_d{dim}_k{k_true}_c{copy_num}", chunk_id=1000 + 17 * i, copy_num=copy_num, band=10,
                        coverage=FEATURE_COVERAGE, local_coverage=n / copy_num,
                        variants=[fbits(row) for row in x], variant_type=[[int(h), int(t)] for h, t in vt]))
    return out


def pileups():
    g = np.load(os.path.join(os.path.dirname(HERE), "small_pileups.npz"))
    frac = float(g["band_frac"][0])
    out = []
    for c in g["chunks"]:
        t0, tl, r0, n = int(c["tmpl_off"]), int(c["tmpl_len"]), int(c["read_first"]), int(c["n_reads"])
        reads, ops, strands = [], [], []
        for r in range(r0, r0 + n):
            reads.append(bytes(g["read_bases"][g["read_off"][r]:g["read_off"][r + 1]]).decode())
            ops.append("".join(OPS[o] for o in g["ops"][g["ops_off"][r]:g["ops_off"][r + 1]]))
            strands.append(int(g["strand"][r]))
        out.append(dict(chunk_id=int(c["chunk_id"]), copy_num=int(c["copy_num"]), coverage=float(g["haploid_coverage"][0]),
                        band_width=int(np.ceil(tl * frac)),  # ReadType::band_width, definitions/src/lib.rs:201-210
                        template=bytes(g["tmpl_bases"][t0:t0 + tl]).decode(), reads=reads, ops=ops, strands=strands))
    return out


def laplacians():
    """normalised graph Laplacians I - D^-1/2 W D^-1/2 (phmm_likelihood_correction.rs:385-402) of two-block similarity
    matrices with 1e-16 links, the shape filter_similarity (:337-356) produces"""
    rng = np.random.default_rng(7)
    out = []
    for n, split in ((12, 6), (30, 11)):
        w = np.full((n, n), 1e-16)
        for a, b in ((0, split), (split, n)):
            blk = rng.uniform(0.55, 0.99, (b - a, b - a))
            w[a:b, a:b] = (blk + blk.T) / 2
        w[0, n - 1] = w[n - 1, 0] = 0.6  # one link across
        rowsum = w.sum(axis=1)
        sq = np.sqrt(1.0 / rowsum)
        lap = -w * sq[:, None] * sq[None, :]
        np.fill_diagonal(lap, 1.0)
        out.append([fbits(row) for row in lap])
    return out


def main():
    inputs = dict(
        format=1,
        hmm=dict(forward=hmm_dict(), reverse=hmm_dict()),
        rng=dict(seeds=[0, 1, 3490, 3490 * 104, 20260101], ranges=[2, 3, 4, 7, 12, 24, 60, 160, 1023],
                 bools=[0.5, 0.25, 0.9, 1e-3, 0.9999999], choose_k=[2, 3, 4, 5, 7], slice_len=[1, 12, 60, 160],
                 weights=[fbits(w) for w in ([1.0, 2.0, 3.0], [0.0, 0.5, 0.0, 0.25], [4.5, 0.0, 0.0, 1e-9, 7.25, 3.0],
                                             np.random.default_rng(3).uniform(0, 9, 60))]),
        features=feature_problems(), pileups=pileups(), eigen=laplacians())
    path = os.environ.get("JTK_REF_INPUTS_OUT", os.path.join(HERE, "inputs.json"))
    with open(path, "w") as f:
        json.dump(inputs, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
