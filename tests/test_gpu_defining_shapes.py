"""GPU parity at the defining shapes, in numbers that exercise the kernels' rare paths (band moves in runs, block boundaries,
generic groups, flushes): the committed 64-chunk golden of the headline workload (no oracle in the loop), 256 chunks of it
against the oracle, Poisson(60) coverage, and one call that mixes the three pair-HMM kernels."""
import os
import sys

import numpy as np
import pytest

import helpers
import oracle_ffi as O
from jtk_amd import api, batch as jb, ffi, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def check_equal(dev, ora, b, tol=1e-4):
    assert np.array_equal(dev["result"]["status"], ora["result"]["status"])
    assert np.array_equal(dev["result"]["polish_rounds"], ora["result"]["polish_rounds"])
    assert np.array_equal(dev["result"]["n_variants"], ora["result"]["n_variants"])
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert np.array_equal(dev["label"], ora["label"])
    assert np.abs(dev["log_post"] - ora["log_post"]).max() < tol
    assert np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"]))       # in fact bit for bit
    assert np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(ora["result"]["score"]))
    n = int(ora["cons_off"][-1])
    assert np.array_equal(dev["cons_off"], ora["cons_off"]) and bytes(dev["cons"][:n]) == bytes(ora["cons"][:n])


def test_cfg3_golden_64_chunks(jtk_lib):
    """tests/golden/cfg3_64.npz (made by the oracle, tests/golden/make_cfg3_64.py): the device reproduces it with no oracle
    in the loop -- a regression in a kernel shows here even where the oracle cannot be built"""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_cfg3_64
    g = np.load(os.path.join(HERE, "golden", "cfg3_64.npz"))
    b, cfg = synth.make_batch("ont_diploid", 64)
    assert make_cfg3_64.inputs_digest(b) == str(g["inputs_sha256"][0]), "the generator no longer produces the golden's inputs"
    p = jb.default_params(cfg["coverage"], cfg["band_frac"])
    dev = api.cluster_chunks(p, b)
    check_equal(dev, dict(result=g["result"], label=g["label"], log_post=g["log_post"], cons=g["cons"], cons_off=g["cons_off"]), b)


def test_cfg3_refit_golden_32_chunks(jtk_lib):
    """tests/golden/cfg3_refit_32.npz (made by the oracle, tests/golden/make_cfg3_refit_32.py): the stage as JTK enters it
    (mod.rs:56-83) at full shape -- model refitted on the stage's training pile-ups, gains calibrated on it -- with no oracle in the
    loop.  The refitted model lets weak columns through the filter: eight of the 32 chunks are ones whose chains accept 10^5 .. 10^6
    moves (bench.REFIT_EVENTFUL), the regime a table-driven walk is most likely to get wrong.  The device's own refit + gains
    calibration must reproduce the stored parameters bit for bit, and the clustering on them the stored results."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_cfg3_refit_32 as mk
    g = np.load(os.path.join(HERE, "golden", "cfg3_refit_32.npz"))
    b, cfg = mk.make_inputs()
    assert mk.inputs_digest(b) == str(g["inputs_sha256"][0]), "the generator no longer produces the golden's inputs"
    p0 = jb.default_params(cfg["coverage"], cfg["band_frac"])
    f, r = api.fit_model(p0, b.subset(mk.training_chunks(b)), rounds=10)
    p = ffi.Params.from_buffer_copy(bytes(p0))
    p.forward, p.reverse = f, r
    p.gains = api.estimate_gains(f, r)
    assert bytes(p) == g["params"].tobytes(), "refit / gains calibration differ from the golden's parameters"
    dev = api.cluster_chunks(p, b)
    check_equal(dev, dict(result=g["result"], label=g["label"], log_post=g["log_post"], cons=g["cons"], cons_off=g["cons_off"]), b)
    assert (g["result"]["n_variants"] >= 1).sum() >= 16   # the refitted model does let columns through


def test_cfg5_golden_32_chunks(jtk_lib):
    """tests/golden/cfg5_32.npz (made by the oracle, tests/golden/make_cfg5_32.py): BASELINE cfg 5 -- HiFi reads, band radius 10,
    the configuration that runs on phmm_pair_kernel (two reads per wave) -- with no oracle in the loop: labels, k, score and
    posterior bits, consensus AND the re-threaded ops (the next round's band follows them)"""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_cfg5_32 as mk
    g = np.load(os.path.join(HERE, "golden", "cfg5_32.npz"))
    b, cfg, p = mk.make_inputs()
    assert mk.inputs_digest(b) == str(g["inputs_sha256"][0]), "the generator no longer produces the golden's inputs"
    assert int(b.chunks["n_reads"][0]) == 40 and p.band_frac == 0.01
    dev = api.cluster_chunks(p, b)
    check_equal(dev, dict(result=g["result"], label=g["label"], log_post=g["log_post"], cons=g["cons"], cons_off=g["cons_off"]), b)
    m = int(g["ops_out_off"][-1])
    assert np.array_equal(dev["ops_out_off"], g["ops_out_off"]) and np.array_equal(dev["ops_out"][:m], g["ops_out"])
    assert (g["result"]["polish_rounds"] >= 2).sum() >= 8   # templates that were edited, i.e. bands that moved


def test_32_more_chunks_of_cfg5_match_the_oracle(jtk_lib, oracle):
    """32 other cfg-5 pile-ups with the live oracle in the loop (RNG streams, error patterns and band paths the golden does not hold)"""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_cfg5_32 as mk
    b, cfg, p = mk.make_inputs(first=5400, count=32)
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert ora["rc"] == 0
    check_equal(dev, ora, b)
    m = int(ora["ops_out_off"][-1])
    assert np.array_equal(dev["ops_out_off"], ora["ops_out_off"]) and np.array_equal(dev["ops_out"][:m], ora["ops_out"][:m])


def test_refit_stage_matches_the_oracle(jtk_lib, oracle):
    """the same stage (refit + gains + clustering) with the oracle in the loop, on sixteen other chunks of cfg 3: RNG streams and
    band paths the golden does not hold; parameters from the device's refit, oracle run on exactly those"""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_cfg3_refit_32 as mk
    g = np.load(os.path.join(HERE, "golden", "cfg3_refit_32.npz"))
    cfg = dict(synth.CONFIGS["ont_diploid"])
    b = jb.pack([synth.make_pileup(c, cfg) for c in range(40, 56)])
    p = ffi.Params.from_buffer_copy(g["params"].tobytes())   # (the refit itself is pinned by the golden test above)
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert ora["rc"] == 0
    check_equal(dev, ora, b)


def test_256_chunks_of_the_headline_workload_match_the_oracle(jtk_lib, oracle):
    """what bench.py's cpu_baseline leg compares on the side, as a test: chunks 64 .. 319 of cfg 3 (other RNG streams and
    band paths than the golden's)"""
    b, cfg = synth.make_batch("ont_diploid", 256, first_chunk_id=64)
    p = jb.default_params(cfg["coverage"], cfg["band_frac"])
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert ora["rc"] == 0
    check_equal(dev, ora, b)


def test_poisson_coverage_matches_the_oracle(jtk_lib, oracle):
    """a real 60x data set has Poisson(60) reads per pile-up: a third of them exceed the 63 reads of the one-register chain"""
    rng = np.random.default_rng(5)
    base = dict(synth.CONFIGS["ont_diploid"])
    piles = []
    for c in range(24):
        cfg = dict(base)
        cfg["reads_per_hap"] = max(2, int(rng.poisson(60)) // 2)
        piles.append(synth.make_pileup(7000 + c, cfg, synth.SEED0, 0))
    b = jb.pack(piles)
    assert (b.chunks["n_reads"] > 63).any() and (b.chunks["n_reads"] < 60).any()
    p = jb.default_params(base["coverage"], base["band_frac"])
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert ora["rc"] == 0
    check_equal(dev, ora, b)


@pytest.mark.parametrize("config,lens", [("ont_diploid", (700, 2000, 4200, 1500, 900)), ("hifi_diploid", (2000, 2600, 1200)),
                                          ("ont_noisy", (2000, 1100))])
def test_one_call_mixing_the_pair_hmm_kernels_matches_the_oracle(jtk_lib, oracle, config, lens):
    """template lengths that put the chunks of ONE call on phmm_pair_kernel (radius <= 14), phmm_kernel (15 .. 30) and
    phmm_wide_kernel (> 30) side by side; HiFi (radius 10 .. 13) and the 5 %-error stress profile too"""
    base = dict(synth.CONFIGS[config])
    piles = []
    for c, L in enumerate(lens):
        cfg = dict(base, tmpl_len=L, reads_per_hap=12)
        piles.append(synth.make_pileup(9000 + 10 * c, cfg, synth.SEED0, 1))
    b = jb.pack(piles)
    p = jb.default_params(12.0, base["band_frac"])
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert ora["rc"] == 0
    check_equal(dev, ora, b)


# ---- BASELINE cfg 4 (4 copies x 40 reads x 2 kbp): the configuration where the K-way table chain (mcmc_chain_tab<K>) is the
#      whole cost.  Round 3 covered it with two chunks.

def test_cfg4_golden_16_chunks(jtk_lib):
    """tests/golden/cfg4_16.npz (made by the oracle, tests/golden/make_cfg4_16.py): sixteen 160-read pile-ups through K = 2, 3, 4
    with no oracle in the loop"""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_cfg3_64
    import make_cfg4_16
    g = np.load(os.path.join(HERE, "golden", "cfg4_16.npz"))
    b, cfg, p = make_cfg4_16.make_inputs()
    assert make_cfg3_64.inputs_digest(b) == str(g["inputs_sha256"][0]), "the generator no longer produces the golden's inputs"
    assert int(b.chunks["n_reads"].min()) == 160 and int(b.chunks["copy_num"][0]) == 4
    dev = api.cluster_chunks(p, b)
    check_equal(dev, dict(result=g["result"], label=g["label"], log_post=g["log_post"], cons=g["cons"], cons_off=g["cons_off"]), b)
    assert g["result"]["cluster_num"].max() >= 3


def test_16_chunks_of_cfg4_match_the_oracle(jtk_lib, oracle):
    """sixteen more (chunk ids 4300 .. 4315: other RNG streams than the golden's) against the oracle run on the box's CPUs"""
    b, cfg = synth.make_batch("ont_4copy", 16, first_chunk_id=4300, min_variants=2)
    p = jb.default_params(cfg["coverage"], cfg["band_frac"])
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert ora["rc"] == 0
    check_equal(dev, ora, b)


def test_copy_numbers_3_5_6_7_at_40_reads_per_copy_match_the_oracle(jtk_lib, oracle):
    """every K the chain kernel is instantiated for beyond the diploid one, at 40 reads per copy (120 .. 280 reads: the
    register tables, and from 256 reads the LDS tables of mcmc_chain_tab), one pile-up each in ONE call"""
    base = dict(synth.CONFIGS["ont_4copy"])
    piles = []
    for i, k in enumerate((3, 5, 6, 7)):
        cfg = dict(base, n_haps=k, copy_num=k, tmpl_len=900, divergence=4e-3)
        piles.append(synth.make_pileup(8800 + 7 * i, cfg, synth.SEED0, 2))
    b = jb.pack(piles)
    assert b.chunks["n_reads"].tolist() == [120, 200, 240, 280]
    p = jb.default_params(base["coverage"], base["band_frac"])
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert ora["rc"] == 0
    check_equal(dev, ora, b)
    assert ora["result"]["cluster_num"].max() >= 3


def test_parity_campaign_seed_31_four_copy_slice(jtk_lib, oracle):
    """scripts/parity_campaign.sh, seed 31, its four-copy leg (scripts/parity_headline.py 12 33 ont_4copy) cut to six random
    chunk ids: promoted from a builder-run log to a test"""
    rng = np.random.default_rng(33)
    ids = [int(x) for x in rng.integers(0, 1 << 40, 6)]
    base = dict(synth.CONFIGS["ont_4copy"])
    b = jb.pack([synth.make_pileup(i, base, synth.SEED0, 1) for i in ids])
    p = jb.default_params(base["coverage"], base["band_frac"])
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert ora["rc"] == 0
    check_equal(dev, ora, b)
