"""The oracle's Metropolis clustering against the reference's own brute-force comparator
(haplotyper/src/local_clustering/exact_clustering.rs:7-26, driven next to cluster_filtered_variants by
sandbox/src/bin/benchmark_mcmc.rs:111-122) -- a reference-derived check of the chain's objective (get_lk
pseudo_mcmc.rs:785-795): for the chain's partition P and the columns it uses, score = sum_c sum_{used d} max(sum_{i in c}
x_id, 0) <= sum_i max_c sum_{d in sel_c} x_id <= the exact optimum, with equality on cleanly separated pile-ups."""
import ctypes as C

import numpy as np
import pytest

import helpers
import oracle_ffi as O
from jtk_amd import batch as jb, ffi


def exact(x, copy_num):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, dim = x.shape
    asn = np.zeros(n, dtype=np.uintp)
    gain = np.zeros((n, copy_num))
    score = O.lib().jo_cluster_filtered_variants_exact(O.f64p(x), n, dim, copy_num, O.szp(asn), O.f64p(gain))
    return score, asn, gain


def oracle_features(p, x, vt, copy_num, chunk_id=1000):
    n, dim = x.shape
    ch = np.zeros(1, dtype=ffi.FEATURE_CHUNK_DT)
    ch[0] = (chunk_id, copy_num, n, dim, 0, 0, 0, 0, n / copy_num)
    lab = np.zeros(n, np.uint32)
    post = np.zeros((n, copy_num))
    res = np.zeros(1, dtype=ffi.RESULT_DT)
    var = np.ascontiguousarray(x.ravel())
    vts = np.ascontiguousarray(vt.ravel().astype(np.uint32))
    rc = O.lib().jo_cluster_features(C.byref(helpers.oracle_params(p)), 1, ch.ctypes.data, O.f64p(var), O.u32p(vts),
                                     O.u32p(lab), O.f64p(post), copy_num, res.ctypes.data, 1)
    return rc, lab, post, res[0]


def test_exact_known_answers(oracle):
    # two columns, each owned by one pair of reads: best = {col1} / {col0} -> 3+3+2+2
    score, asn, gain = exact([[3, -2], [3, -2], [-3, 2], [-3, 2]], 2)
    assert score == 10.0
    assert asn.tolist() == [1, 1, 0, 0]                     # selections are kept non-increasing: (0b10, 0b01)
    assert gain.tolist() == [[-2, 3], [-2, 3], [2, -3], [2, -3]]
    # nothing to gain: the all-empty selection (score 0) stays the arg-max, every read in the LAST maximum
    score, asn, gain = exact([[-1, -2], [-3, -1]], 2)
    assert score == 0.0 and asn.tolist() == [1, 1] and not gain.any()
    # one cluster: the best single selection
    score, asn, _ = exact([[1, -4, 2], [2, 1, -1], [1, -1, 0.5]], 1)
    assert score == 5.5 and asn.tolist() == [0, 0, 0]      # {col0, col2}: 3 + 1 + 1.5
    # the candidate that selects every column in every cluster is never scored (`while != last_loop`, :16)
    score, _, _ = exact([[1.0]], 1)
    assert score == 0.0


@pytest.mark.parametrize("seed", range(6))
def test_chain_score_never_exceeds_the_exact_optimum(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    p = jb.default_params(haploid_coverage=6.0)
    for n, dim, k_true, copy_num in [(12, 3, 2, 2), (10, 4, 2, 2), (12, 2, 3, 3), (9, 3, 1, 2), (11, 5, 2, 2)]:
        x, vt, _ = helpers.random_feature_problem(rng, n, dim, k_true)
        rc, lab, post, res = oracle_features(p, x, vt, copy_num, chunk_id=seed * 31 + n)
        assert rc == 0 and res["status"] == 0
        best, _, _ = exact(x, copy_num)
        assert res["score"] <= best + 1e-9, (n, dim, res["score"], best)


def test_chain_reaches_the_exact_optimum_on_clean_pileups(oracle):
    """two well separated haplotypes, every column informative: the chain's best partition is the exact one"""
    rng = np.random.default_rng(5)
    p = jb.default_params(haploid_coverage=6.0)
    hits = 0
    for trial in range(8):
        n, dim = 12, 3
        lab = np.array([0] * 6 + [1] * 6)
        owner = np.array([0, 1, 0])
        x = np.where(lab[:, None] == owner[None, :], rng.normal(5.0, 0.3, (n, dim)), rng.normal(-5.0, 0.3, (n, dim)))
        vt = np.stack([np.ones(dim), np.zeros(dim)], axis=1).astype(np.uint32)
        rc, got, post, res = oracle_features(p, x, vt, 2, chunk_id=trial)
        best, asn, _ = exact(x, 2)
        assert rc == 0 and res["cluster_num"] == 2
        assert res["score"] <= best + 1e-9
        assert helpers.same_partition(got, lab) and helpers.same_partition(asn, lab)
        hits += abs(res["score"] - best) < 1e-9
    assert hits == 8


def test_oracle_fails_where_the_reference_asserts(oracle):
    """LKCount::add's zero band (pseudo_mcmc.rs:830): a value of exactly +-POS_THR is neither positive, negative nor
    `abs() < POS_THR` -> the reference panics; so does a NaN size table (:714-715).  The oracle reports the chunk."""
    rng = np.random.default_rng(9)
    p = jb.default_params(haploid_coverage=6.0)
    x, vt, _ = helpers.random_feature_problem(rng, 12, 3, 2)
    rc, _, _, res = oracle_features(p, x, vt, 2)
    assert rc == 0 and res["status"] == 0
    for bad in (1e-5, -1e-5, float("nan")):
        y = x.copy()
        y[7, 1] = bad
        rc, _, _, res = oracle_features(p, y, vt, 2)
        assert rc != 0 and res["status"] == -6, bad
    y = x.copy()
    y[7, 1] = 0.99e-5                                   # inside the band: counted as zero, no panic
    rc, _, _, res = oracle_features(p, y, vt, 2)
    assert rc == 0 and res["status"] == 0
    for cov in (float("nan"), 0.0, -3.0, float("inf")):  # size_to_lk[0] = 0 ln(lambda) - lambda is NaN
        p.haploid_coverage = cov
        rc, _, _, res = oracle_features(p, x, vt, 2)
        assert rc != 0 and res["status"] == -6, cov
    p.haploid_coverage = 1e-300                          # tiny but positive: a valid table
    rc, _, _, res = oracle_features(p, x, vt, 2)
    assert rc == 0 and res["status"] == 0
