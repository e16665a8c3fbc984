"""CPU checks of the oracle's clustering_recursive restatement (mod.rs:125-189): the properties the reference's
own code asserts (posteriors of the merged clustering sum to one, mod.rs:184; copy numbers add up, :241) and
recovery of the generating haplotypes on a synthetic 8-copy pile-up."""
import numpy as np

import helpers
import oracle_ffi as O


def test_split_branch_recovers_eight_copies_and_normalises_posteriors():
    b, cfg, p = helpers.small_batch(config="ont_4copy", n_chunks=2, tmpl_len=600, reads_per_hap=10, n_haps=8,
                                    copy_num=9, divergence=2e-2, min_variants=3)
    out = O.cluster_chunks(helpers.oracle_params(p), b, n_threads=2)
    assert out["rc"] == 0
    ks = out["result"]["cluster_num"]
    assert ks.max() > 4 and ks.max() <= 9          # more clusters than one 4-way pass can give, at most copy_num
    for c in range(b.n_chunks):
        reads = list(b.chunk_reads(c))
        k = int(ks[c])
        lab = out["label"][reads]
        assert lab.max() < k
        rows = out["log_post"][reads][:, :k]
        assert np.abs(np.exp(rows).sum(axis=1) - 1.0).max() < 1e-4
    reads = list(b.chunk_reads(0))
    lab, truth = out["label"][reads], b.truth[reads]
    pure = sum(np.bincount(truth[lab == c]).max() for c in range(int(ks[0])))
    assert ks[0] == 8 and pure >= 0.95 * len(reads)   # eight clusters, each (almost) one generating copy


def test_split_branch_returns_the_single_cluster_of_the_first_pass():
    b, cfg, p = helpers.small_batch(n_chunks=1, tmpl_len=300, reads_per_hap=3)
    b.chunks["copy_num"][0] = 8                    # diploid pile-up of 6 reads declared 8-copy: n <= 4 is false, k stays 1
    out = O.cluster_chunks(helpers.oracle_params(p), b)
    assert out["rc"] == 0 and out["result"]["cluster_num"][0] == 1
    assert not out["label"].any() and not out["log_post"].any()
