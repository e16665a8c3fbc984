"""oracle/correction.c: the restatement of phmm_likelihood_correction.rs:32-97 behaves like the reference's description of
it (CPU only; parity with nalgebra / rand_xoshiro is unpinned, see the file header)."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from helpers import correction_problem, same_partition
from jtk_amd import ffi


def run(prob, selection=None, cov=20.0, min_gain=1.0, want_sims=0):
    chunks = prob["chunks"].copy()
    sel = np.arange(len(chunks), dtype=np.uint64) if selection is None else np.asarray(selection, dtype=np.uint64)
    rc, cluster, touched, ari, sims = O.correct_clustering(prob["read_id"], prob["node_off"], prob["nodes"], prob["posteriors"],
                                                           chunks, sel, cov, min_gain, want_sims)
    return rc, cluster, touched, ari, sims, chunks


def test_xoroshiro128pp_reference_vector():
    """Xoroshiro128++ with state (1, 2): the known-answer vector of the reference implementation (Blackman & Vigna's
    xoroshiro128plusplus.c, the vector rand_xoshiro 0.6.0 tests its Xoroshiro128PlusPlus against), plus an independent
    evaluation of the published recurrence in Python integers."""
    published = [393217, 669327710093319, 1732421326133921491, 11394790081659126983, 9555452776773192676,
                 3586421180005889563, 1691397964866707553, 10735626796753111697, 15216282715349408991,
                 14247243556711267923]
    M = (1 << 64) - 1

    def rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & M

    s0, s1 = 1, 2
    want = []
    for _ in range(10):
        want.append((rotl((s0 + s1) & M, 17) + s0) & M)
        s1 ^= s0
        s0, s1 = rotl(s0, 49) ^ s1 ^ ((s1 << 21) & M), rotl(s1, 28)
    assert want == published
    r = O.Rng()
    r.s[0], r.s[1], r.kind = 1, 2, 1
    got = [O.lib().jo_rng_next_u64(C.byref(r)) for _ in range(10)]
    assert got == published
    # seed_from_u64: two SplitMix64 outputs; SplitMix64(0) starts 0xe220a8397b1dcdaf, 0x6e789e6aa1b965f4 (public vector)
    O.lib().jo_rng128pp_seed_from_u64(C.byref(r), 0)
    assert (r.s[0], r.s[1]) == (0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4)


def test_similarity_matrix_properties():
    prob = correction_problem(3, n_chunks=5, n_reads=30)
    n0 = int((prob["nodes"]["chunk"] == 0).sum())
    rc, cluster, touched, ari, sims, chunks = run(prob, want_sims=n0)
    assert rc == 0
    assert np.all(np.diag(sims) == 0.0)
    off = sims[~np.eye(n0, dtype=bool)]
    assert np.all((off > 0.0) & (off < 1.0))
    # alignment() is symmetric in its two contexts up to the order of the max3 / fmax arguments, which compare equal values
    assert np.array_equal(sims, sims.T)


def test_correction_repairs_flat_posteriors():
    """reads of one haplotype end in one cluster on every corrected chunk, including the nodes whose own posterior is flat"""
    prob = correction_problem(11, n_chunks=6, n_reads=60, flat=0.2)
    rc, cluster, touched, ari, sims, chunks = run(prob, min_gain=1e9)  # nothing protected
    assert rc == 0
    nodes, hap = prob["nodes"], prob["hap"]
    read_of = np.repeat(np.arange(len(hap)), np.diff(prob["node_off"]).astype(int))
    n_split = 0
    for c in range(len(chunks)):
        m = nodes["chunk"] == c
        assert touched[m].all() or not touched[m].any()
        if chunks["cluster_num"][c] == 2 and touched[m].any():
            n_split += 1
            agree = (cluster[m] == hap[read_of[m]]).mean()
            assert max(agree, 1.0 - agree) >= 0.9, (c, agree)
    assert n_split >= 3
    # untouched nodes keep their labels
    assert np.array_equal(cluster[touched == 0], nodes["cluster"][touched == 0])


def test_selection_and_single_cluster_chunks():
    prob = correction_problem(5, n_chunks=6, n_reads=40, single=(2,))
    rc, cluster, touched, ari, sims, chunks = run(prob, selection=[0, 2, 4], min_gain=1e9)
    assert rc == 0
    nodes = prob["nodes"]
    assert not touched[np.isin(nodes["chunk"], [1, 2, 3, 5])].any()  # unselected, or a single cluster (:40)
    assert touched[np.isin(nodes["chunk"], [0, 4])].all()
    assert np.isnan(ari[[1, 2, 3, 5]]).all() and not np.isnan(ari[[0, 4]]).any()


def test_protection_keeps_suppressed_chunks():
    """a chunk whose correction would be suppressed keeps its clustering when its local-clustering score protects it (:57-59)"""
    prob = correction_problem(7, n_chunks=6, n_reads=50, wrong=0.05)
    rc0, cl0, t0, ari0, _, ch0 = run(prob, min_gain=1e9)
    rc1, cl1, t1, ari1, _, ch1 = run(prob, min_gain=0.0)  # everything with a positive score is protected
    assert rc0 == 0 and rc1 == 0
    assert np.array_equal(ari0, ari1, equal_nan=True)
    supp0 = [c for c in range(6) if ch0["cluster_num"][c] == 1]
    assert len(supp0) >= 1  # the lowest 5 % of the ARIs sit below the threshold: at least the minimum is suppressed
    for c in supp0:
        m = prob["nodes"]["chunk"] == c
        assert ch1["cluster_num"][c] == 2 and not t1[m].any()


def test_reference_panics_are_reported():
    prob = correction_problem(2, n_chunks=4, n_reads=3, window=(4, 4))  # 3 reads: n - n / copy_num / 4 == n -> sims[pivot] :361
    rc, *_ = run(prob)
    assert rc == -6
    prob = correction_problem(2, n_chunks=4, n_reads=30)
    prob["nodes"]["chunk"][5] = 99  # beyond the largest chunk id: obs_counts[chunk] :141
    rc, *_ = run(prob)
    assert rc == -6
    prob = correction_problem(2, n_chunks=4, n_reads=30)
    prob["posteriors"][0] = 0.5  # a log-probability sum above 0: logit_from_lnp asserts :565
    prob["posteriors"][1] = 0.5
    rc, *_ = run(prob)
    assert rc == -6


def test_estimate_minimum_gain_properties():
    """likelihood_gains.rs:6-39: deterministic in its seed, floored at 1, and for the default ONT-like model a deleted base
    costs a read a few nats (the third smallest of the per-template medians)"""
    from jtk_amd import batch as jb
    from helpers import oracle_params
    p = oracle_params(jb.default_params(haploid_coverage=30.0))
    f = lambda seed, n=12: O.lib().jo_estimate_minimum_gain(C.byref(p.forward), C.byref(p.reverse), seed, n, 40, 100, 25, 8)
    a, b = f(23908), f(23908)
    assert a == b and a >= 1.0
    assert 1.0 <= a < 20.0
    assert np.isnan(O.lib().jo_estimate_minimum_gain(C.byref(p.forward), C.byref(p.reverse), 1, 2, 40, 100, 25, 1))
