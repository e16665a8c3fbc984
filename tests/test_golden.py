"""Committed golden vectors (tests/golden/small_pileups.npz, produced by tests/golden/make_golden.py with the CPU
oracle): the oracle must keep reproducing them (CPU), and the HIP path must reproduce them through the C-ABI (GPU)."""
import os

import numpy as np
import pytest

import helpers
import oracle_ffi as O
from jtk_amd import api, batch as jb

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "small_pileups.npz")


def load():
    g = np.load(GOLD)
    b = jb.Batch(chunks=g["chunks"], tmpl_bases=g["tmpl_bases"], read_bases=g["read_bases"], read_off=g["read_off"],
                 ops=g["ops"], ops_off=g["ops_off"], strand=g["strand"])
    p = jb.default_params(haploid_coverage=float(g["haploid_coverage"][0]), band_frac=float(g["band_frac"][0]))
    return g, b, p


def check(out, g, prefix=""):
    assert np.array_equal(out["label"], g[prefix + "label"])
    assert np.array_equal(out["result"]["cluster_num"], g[prefix + "result"]["cluster_num"])
    assert np.array_equal(out["result"]["n_variants"], g[prefix + "result"]["n_variants"])
    assert np.abs(out["log_post"] - g[prefix + "log_post"]).max() < 1e-4
    assert np.abs(out["result"]["score"] - g[prefix + "result"]["score"]).max() < 1e-4


def test_oracle_reproduces_golden(oracle):
    g, b, p = load()
    out = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    check(out, g)
    n = int(g["cons_off"][-1])
    assert bytes(out["cons"][:n]) == bytes(g["cons"])
    assert np.array_equal(out["result"]["polish_rounds"], g["result"]["polish_rounds"])
    check(O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=True), g, "polished_")


@pytest.mark.gpu
def test_gpu_reproduces_golden(jtk_lib):
    assert jtk_lib.jtk_lc_device_ok(0) == 1
    g, b, p = load()
    out = api.cluster_chunks(p, b)
    check(out, g)
    n, m = int(g["cons_off"][-1]), int(g["ops_out_off"][-1])
    assert np.array_equal(out["cons_off"], g["cons_off"]) and bytes(out["cons"][:n]) == bytes(g["cons"])
    assert np.array_equal(out["ops_out_off"], g["ops_out_off"]) and np.array_equal(out["ops_out"][:m], g["ops_out"])
    assert np.array_equal(out["result"]["polish_rounds"], g["result"]["polish_rounds"])
    check(api.cluster_polished(p, b), g, "polished_")
