"""jtk_lc_correct_clustering (device similarity fill + host spectral clustering) against oracle/correction.c:
similarity matrices bit for bit, labels / cluster_num / touched flags equal (phmm_likelihood_correction.rs:32-97)."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from helpers import bits, correction_problem
from jtk_amd import api, ffi

pytestmark = pytest.mark.gpu


def both(prob, selection=None, cov=20.0, min_gain=1e9, want_sims=0):
    sel = np.arange(len(prob["chunks"]), dtype=np.uint64) if selection is None else np.asarray(selection, dtype=np.uint64)
    och = prob["chunks"].copy()
    orc, ocl, otouched, _, osims = O.correct_clustering(prob["read_id"], prob["node_off"], prob["nodes"], prob["posteriors"], och,
                                                        sel, cov, min_gain, want_sims)
    gch = prob["chunks"].copy()
    L = ffi.lib()
    L.jtk_lc_debug_cc_keep_sims(1 if want_sims else 0)
    try:
        gcl, gtouched = api.correct_clustering(prob["read_id"], prob["node_off"], prob["nodes"], prob["posteriors"], gch, sel, cov,
                                               min_gain)
        grc = 0
    except ffi.JtkError as e:
        grc, gcl, gtouched = e.status, None, None
    gsims = None
    if want_sims and grc == 0:
        gsims = np.zeros((want_sims, want_sims))
        L.jtk_lc_debug_cc_first_sims.restype = C.c_size_t
        L.jtk_lc_debug_cc_first_sims.argtypes = [C.POINTER(C.c_double), C.c_size_t]
        assert L.jtk_lc_debug_cc_first_sims(ffi.f64p(gsims), gsims.size) == gsims.size
    L.jtk_lc_debug_cc_keep_sims(0)
    return (orc, ocl, otouched, och, osims), (grc, gcl, gtouched, gch, gsims)


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(wrong=0.05)), (3, dict(n_chunks=10, n_reads=120, window=(2, 9), wrong=0.03)),
                                     (4, dict(single=(1, 4), n_chunks=7)), (5, dict(flat=0.5, noise=0.5, wrong=0.1))])
def test_correction_matches_oracle(seed, kw):
    prob = correction_problem(seed, **kw)
    first = next(c for c in range(len(prob["chunks"])) if prob["chunks"]["cluster_num"][c] > 1)
    n0 = int((prob["nodes"]["chunk"] == first).sum())
    (orc, ocl, ot, och, osims), (grc, gcl, gt, gch, gsims) = both(prob, want_sims=n0)
    assert orc == 0 and grc == 0
    assert np.array_equal(bits(osims), bits(gsims))
    assert np.array_equal(ot, gt)
    assert np.array_equal(ocl, gcl)
    assert np.array_equal(och, gch)


@pytest.mark.changes_env
def test_correction_in_several_device_batches_matches_oracle(monkeypatch):
    """jtk_lc_correct_clustering runs its jobs through the device in batches bounded by the bytes of their similarity matrices
    (a genome-scale DataSet must not need all of them at once): with a budget of one matrix per batch the result is unchanged"""
    import os
    prob = correction_problem(3, n_chunks=10, n_reads=120, window=(2, 9), wrong=0.03)
    monkeypatch.setenv("JTK_CC_SIMS_BUDGET", "100")   # fewer doubles than any matrix: every job is its own batch
    (orc, ocl, ot, och, _), (grc, gcl, gt, gch, _) = both(prob)
    assert orc == 0 and grc == 0
    assert np.array_equal(ot, gt) and np.array_equal(ocl, gcl) and np.array_equal(och, gch)


def test_correction_protected_and_selected():
    prob = correction_problem(21, n_chunks=8, n_reads=80, wrong=0.05)
    for min_gain in (0.0, 0.2, 1e9):
        (orc, ocl, ot, och, _), (grc, gcl, gt, gch, _) = both(prob, selection=[0, 2, 3, 6], min_gain=min_gain)
        assert orc == 0 and grc == 0
        assert np.array_equal(ot, gt) and np.array_equal(ocl, gcl) and np.array_equal(och, gch)


def test_correction_copy_number_three():
    """three clusters on a copy-number-3 chunk chain: the copy-number estimate (:131-181) feeds sim() on the device"""
    prob = correction_problem(31, n_chunks=5, n_reads=90)
    rng = np.random.default_rng(31)
    # rewrite every node as a 3-cluster posterior
    hap3 = rng.integers(0, 3, len(prob["hap"]))
    read_of = np.repeat(np.arange(len(hap3)), np.diff(prob["node_off"]).astype(int))
    n = len(prob["nodes"])
    z = rng.normal(0.0, 0.7, (n, 3))
    z[np.arange(n), hap3[read_of]] += 3.0
    lp = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    lp = np.minimum(lp, -1e-9)
    prob["posteriors"] = lp.reshape(-1).copy()
    prob["nodes"]["post_len"] = 3
    prob["nodes"]["post_off"] = np.arange(n) * 3
    prob["nodes"]["cluster"] = lp.argmax(axis=1)
    prob["chunks"]["cluster_num"] = 3
    prob["chunks"]["copy_num"] = 3
    n0 = int((prob["nodes"]["chunk"] == 0).sum())
    (orc, ocl, ot, och, osims), (grc, gcl, gt, gch, gsims) = both(prob, cov=30.0, want_sims=n0)
    assert orc == 0 and grc == 0
    assert np.array_equal(bits(osims), bits(gsims))
    assert np.array_equal(ot, gt) and np.array_equal(ocl, gcl) and np.array_equal(och, gch)


def test_correction_panics_match():
    prob = correction_problem(2, n_chunks=4, n_reads=3, window=(4, 4))
    (orc, *_), (grc, *_) = both(prob)
    assert orc == -6 and grc == -6
    prob = correction_problem(2, n_chunks=4, n_reads=30)
    prob["nodes"]["chunk"][5] = 99
    (orc, *_), (grc, *_) = both(prob)
    assert orc == -6 and grc == -6
    prob = correction_problem(2, n_chunks=4, n_reads=30)
    prob["posteriors"][0] = 0.5
    prob["posteriors"][1] = 0.5
    (orc, *_), (grc, *_) = both(prob)
    assert orc == -6 and grc == -6
