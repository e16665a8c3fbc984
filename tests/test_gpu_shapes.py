"""Parity at the DEFINING shapes of the BASELINE.json configurations (the small-shape tests of test_gpu_parity.py take
other code paths: table-driven chain instead of the generic one, one table register instead of two, ...), a fixed-seed
slice of the randomized campaigns of scripts/parity_*.py, the reference's panics as chunk failures, and the
reference's brute-force comparator as an upper bound of the device chain's score.  All through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers
import oracle_ffi as O
from jtk_amd import api, batch as jb, ffi, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def lib(jtk_lib):
    assert os.environ.get("JTK_DEVICE_IS_ORACLE") or jtk_lib.jtk_lc_device_ok(0) == 1, "needs a gfx950 device"
    return jtk_lib


def assert_full_parity(dev, ora, b):
    assert ora["rc"] == 0
    assert np.array_equal(dev["result"]["status"], ora["result"]["status"])
    assert np.array_equal(dev["result"]["polish_rounds"], ora["result"]["polish_rounds"])
    assert np.array_equal(dev["cons_off"], ora["cons_off"])
    n = int(dev["cons_off"][-1])
    assert bytes(dev["cons"][:n]) == bytes(ora["cons"][:n])
    assert np.array_equal(dev["ops_out_off"], ora["ops_out_off"])
    m = int(dev["ops_out_off"][-1])
    assert np.array_equal(dev["ops_out"][:m], ora["ops_out"][:m])
    assert np.array_equal(dev["result"]["n_variants"], ora["result"]["n_variants"])
    assert np.array_equal(dev["label"], ora["label"])                       # bit-exact integer labels
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert np.abs(dev["log_post"] - ora["log_post"]).max() < TOL            # north_star tolerance
    assert np.abs(dev["result"]["score"] - ora["result"]["score"]).max() < TOL
    # this build: identical bits
    assert np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"]))
    assert np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(ora["result"]["score"]))
    for c in range(b.n_chunks):
        k = int(dev["result"][c]["cluster_num"])
        rows = dev["log_post"][list(b.chunk_reads(c))][:, :k]
        assert np.abs(np.log(np.exp(rows).sum(axis=1))).max() < 1e-4       # mod.rs:184-185


def full_shape(config, n_chunks, first, **kw):
    b, cfg = synth.make_batch(config, n_chunks, first_chunk_id=first, **kw)
    cov = cfg["coverage"] if "reads_per_hap" not in kw else float(kw["reads_per_hap"])
    return b, cfg, jb.default_params(haploid_coverage=cov, band_frac=cfg["band_frac"])


def test_cfg4_full_shape_matches_oracle(lib):
    """BASELINE cfg 4 at its defining shape: 4 copies x 40 reads x 2 kbp (160 reads), candidate k = 2, 3, 4 -- every
    pass runs the chain for more than 127 reads, K = 3 and 4 on 160 reads"""
    b, cfg, p = full_shape("ont_4copy", 2, first=4100, min_variants=2)
    assert int(b.chunks["n_reads"][0]) == 160 and int(b.chunks["tmpl_len"][0]) >= 1900
    ora = O.cluster_chunks(helpers.oracle_params(p), b)
    dev = api.cluster_chunks(p, b)
    assert_full_parity(dev, ora, b)
    assert ora["result"]["cluster_num"].max() >= 3, "the inputs must reach k >= 3"


def test_cfg5_full_shape_matches_oracle(lib):
    """BASELINE cfg 5 at its defining shape: HiFi error model, 40 reads x 2 kbp, band radius 10, the whole path"""
    b, cfg, p = full_shape("hifi_diploid", 3, first=5200, min_variants=1)
    assert int(b.chunks["n_reads"][0]) == 40 and p.band_frac == 0.01
    ora = O.cluster_chunks(helpers.oracle_params(p), b)
    dev = api.cluster_chunks(p, b)
    assert_full_parity(dev, ora, b)


@pytest.mark.parametrize("rph", [32, 39, 47])
def test_poisson_tail_pileups_match_oracle(lib, rph):
    """60x ONT pile-ups are Poisson(60) deep in real data: 64..94 reads x 2 kbp take the two-register tables of the
    diploid chain"""
    b, cfg, p = full_shape("ont_diploid", 2, first=6000 + rph, reads_per_hap=rph, min_variants=1)
    assert 64 <= int(b.chunks["n_reads"][0]) <= 94
    ora = O.cluster_chunks(helpers.oracle_params(p), b)
    dev = api.cluster_chunks(p, b)
    assert_full_parity(dev, ora, b)


def test_headline_shape_random_ids_match_oracle(lib):
    """a fixed-seed slice of scripts/parity_headline.py: cfg 2/3 pile-ups (60 reads x 2 kbp) with chunk ids -- RNG
    streams -- no other test uses"""
    first = int(np.random.default_rng(2).integers(0, 1 << 40))
    b, cfg, p = full_shape("ont_diploid", 6, first=first)
    ora = O.cluster_chunks(helpers.oracle_params(p), b)
    dev = api.cluster_chunks(p, b)
    assert_full_parity(dev, ora, b)


def test_random_shape_sweep_matches_oracle(lib):
    """a fixed-seed slice of scripts/parity_sweep_full.py: random configuration, template length, depth and chunk ids;
    every fifth batch goes through clustering_recursive's split"""
    for it, b, p in helpers.shape_sweep_inputs():
        dev = api.cluster_chunks(p, b, raise_on_chunk_failure=False)
        ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
        assert_full_parity(dev, ora, b)


def test_random_chain_sweep_matches_oracle(lib):
    """a fixed-seed slice of scripts/parity_sweep.py: random feature problems, K = 2..4"""
    import test_gpu_parity as T
    for seed in (300, 301):
        rng = np.random.default_rng(seed)
        specs = []
        for _ in range(14):
            cn = int(rng.choice([2, 2, 2, 2, 3, 4]))
            n = int(rng.integers(4, 128)) if cn == 2 else int(rng.integers(6, 90))
            d = int(rng.integers(1, 9)) if cn == 2 else int(rng.integers(1, 3 * cn + 1))
            specs.append((n, d, int(rng.integers(1, cn + 1)), cn))
        p = jb.default_params(haploid_coverage=float(rng.choice([8.0, 15.0, 30.0])))
        dev, ora, _ = T.run_features_both(p, specs, seed=seed)
        assert np.array_equal(dev["label"], ora["label"])
        assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
        assert np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"]))
        assert np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(ora["result"]["score"]))


def one_feature_chunk(x, vt, copy_num, chunk_id=1000):
    n, dim = x.shape
    ch = np.zeros(1, dtype=ffi.FEATURE_CHUNK_DT)
    ch[0] = (chunk_id, copy_num, n, dim, 0, 0, 0, 0, n / copy_num)
    return ch, np.ascontiguousarray(x.ravel()), np.ascontiguousarray(vt.ravel().astype(np.uint32))


def test_chunk_fails_where_the_reference_asserts(lib):
    """LKCount's zero-band assert (pseudo_mcmc.rs:830,:841: a value of exactly +-POS_THR, or NaN) and the NaN check of
    the size table (:714-715) are panics in the reference: the device reports JTK_ERR_CHUNK_FAILED for that chunk,
    exactly where the oracle does, and clusters the other chunks of the batch"""
    rng = np.random.default_rng(9)
    p = jb.default_params(haploid_coverage=6.0)
    x, vt, _ = helpers.random_feature_problem(rng, 12, 3, 2)
    for bad in (1e-5, -1e-5, float("nan"), 0.99e-5):
        y = x.copy()
        y[7, 1] = bad
        ch, var, vts = one_feature_chunk(y, vt, 2)
        out = api.cluster_features(p, ch, var, vts, 2, raise_on_chunk_failure=False)
        expect_fail = bad != 0.99e-5
        assert (out["rc"] == -6 and out["result"]["status"][0] == -6) == expect_fail, bad
        if not expect_fail:
            assert out["rc"] == 0
    ch, var, vts = one_feature_chunk(x, vt, 2)
    for cov in (float("nan"), 0.0, -3.0, float("inf")):
        p.haploid_coverage = cov
        out = api.cluster_features(p, ch, var, vts, 2, raise_on_chunk_failure=False)
        assert out["rc"] == -6 and out["result"]["status"][0] == -6, cov
    # a failing chunk does not take its neighbours with it
    p.haploid_coverage = 6.0
    y = x.copy()
    y[0, 0] = 1e-5
    chunks = np.zeros(2, dtype=ffi.FEATURE_CHUNK_DT)
    chunks[0] = (1, 2, 12, 3, 0, 0, 0, 0, 6.0)
    chunks[1] = (1000, 2, 12, 3, 0, 36, 3, 12, 6.0)
    var = np.concatenate([y.ravel(), x.ravel()])
    vts = np.concatenate([vt.ravel(), vt.ravel()]).astype(np.uint32)
    out = api.cluster_features(p, chunks, var, vts, 2, raise_on_chunk_failure=False)
    assert out["rc"] == -6 and out["result"]["status"].tolist() == [-6, 0]
    good = api.cluster_features(p, *one_feature_chunk(x, vt, 2), 2)
    assert np.array_equal(out["label"][12:], good["label"])


@pytest.mark.parametrize("seed", range(3))
def test_device_chain_score_is_bounded_by_the_exact_optimum(lib, seed):
    """cluster_filtered_variants_exact (exact_clustering.rs:7-26), the comparator of sandbox benchmark_mcmc.rs:111-122:
    the device chain's score never exceeds it, and reaches it on cleanly separated pile-ups"""
    rng = np.random.default_rng(200 + seed)
    p = jb.default_params(haploid_coverage=6.0)
    for n, dim, k_true, copy_num in [(12, 3, 2, 2), (10, 4, 2, 2), (12, 2, 3, 3), (11, 5, 2, 2)]:
        x, vt, _ = helpers.random_feature_problem(rng, n, dim, k_true)
        out = api.cluster_features(p, *one_feature_chunk(x, vt, copy_num, chunk_id=seed * 31 + n), copy_num)
        asn = np.zeros(n, dtype=np.uintp)
        gain = np.zeros((n, copy_num))
        best = O.lib().jo_cluster_filtered_variants_exact(O.f64p(np.ascontiguousarray(x)), n, dim, copy_num, O.szp(asn),
                                                          O.f64p(gain))
        assert out["result"]["score"][0] <= best + 1e-9
    lab = np.array([0] * 6 + [1] * 6)
    owner = np.array([0, 1, 0])
    x = np.where(lab[:, None] == owner[None, :], rng.normal(5.0, 0.3, (12, 3)), rng.normal(-5.0, 0.3, (12, 3)))
    vt = np.stack([np.ones(3), np.zeros(3)], axis=1).astype(np.uint32)
    out = api.cluster_features(p, *one_feature_chunk(x, vt, 2, chunk_id=seed), 2)
    asn = np.zeros(12, dtype=np.uintp)
    gain = np.zeros((12, 2))
    best = O.lib().jo_cluster_filtered_variants_exact(O.f64p(np.ascontiguousarray(x)), 12, 3, 2, O.szp(asn), O.f64p(gain))
    assert abs(out["result"]["score"][0] - best) < 1e-9
    assert helpers.same_partition(out["label"], lab)


@pytest.mark.parametrize("radius,take_num,ignore_edge", [(0, 0, 3), (20, 8, 0), (0, 5, 0), (12, 0, 0)])
def test_window_polishing_matches_oracle(lib, radius, take_num, ignore_edge):
    """jtk_lc_polish_chunks = polish_until_converge_antidiagonal(.., HMMPolishConfig::new(radius, take_num, ignore_edge)) on
    independent windows: the call consensus::polish_seg makes with (radius / 2, max_coverage, 0) (consensus/mod.rs:476-483);
    only the first take_num reads vote, every read's ops are re-threaded"""
    b, cfg, p = helpers.small_batch(config="ont_noisy", n_chunks=4, tmpl_len=700, reads_per_hap=7, first=900, tmpl_err=8e-3)
    dev = api.polish_chunks(p, b, radius=radius, take_num=take_num, ignore_edge=ignore_edge)
    ora = O.polish_chunks(helpers.oracle_params(p), b, radius=radius, take_num=take_num, ignore_edge=ignore_edge)
    assert ora["rc"] == 0 and dev["rc"] == 0
    assert np.array_equal(dev["result"]["polish_rounds"], ora["result"]["polish_rounds"])
    assert ora["result"]["polish_rounds"].max() >= 2, "the windows must need polishing"
    assert np.array_equal(dev["cons_off"], ora["cons_off"])
    n = int(dev["cons_off"][-1])
    assert bytes(dev["cons"][:n]) == bytes(ora["cons"][:n])
    assert np.array_equal(dev["ops_out_off"], ora["ops_out_off"])
    m = int(dev["ops_out_off"][-1])
    assert np.array_equal(dev["ops_out"][:m], ora["ops_out"][:m])
    if take_num:                      # fewer voters -> (generally) a different consensus than the all-read one
        full = O.polish_chunks(helpers.oracle_params(p), b, radius=radius, take_num=0, ignore_edge=ignore_edge)
        assert full["rc"] == 0


def test_deep_window_polishes_beyond_the_chain_limit(lib):
    """consensus::polish_seg windows can be deep: 1,100 reads on a 180-bp window, 30 of them voting (a polish-only call has no
    chain at all)"""
    b, cfg, p = helpers.small_batch(config="ont_noisy", n_chunks=1, tmpl_len=180, reads_per_hap=550, first=77, tmpl_err=1e-2)
    assert int(b.chunks["n_reads"][0]) == 1100
    dev = api.polish_chunks(p, b, radius=20, take_num=30, ignore_edge=0)
    ora = O.polish_chunks(helpers.oracle_params(p), b, radius=20, take_num=30, ignore_edge=0)
    assert ora["rc"] == 0 and dev["rc"] == 0 and int(dev["result"]["status"][0]) == 0
    assert np.array_equal(dev["result"]["polish_rounds"], ora["result"]["polish_rounds"])
    n = int(dev["cons_off"][-1])
    assert np.array_equal(dev["cons_off"], ora["cons_off"]) and bytes(dev["cons"][:n]) == bytes(ora["cons"][:n])
    m = int(dev["ops_out_off"][-1])
    assert np.array_equal(dev["ops_out_off"], ora["ops_out_off"]) and np.array_equal(dev["ops_out"][:m], ora["ops_out"][:m])


def test_full_path_pileup_of_1100_reads_matches_oracle(lib):
    """the whole stage path on a pile-up deeper than the table-driven chains take (1,100 reads x 300 bp, diploid): polish,
    tables, filter on the ordinary kernels, the chain in mcmc_kernel_huge (session class 2) -- no JTK_ERR_UNSUPPORTED for a read
    count any more"""
    b, p = helpers.full_path_1100_inputs()
    assert int(b.chunks["n_reads"][0]) == 1100
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert_full_parity(dev, ora, b)


# ---- bands wider than one wavefront (phmm_wide_kernel): CLR / None reads, ONT chunks longer than 2,033 bp

@pytest.mark.parametrize("tmpl_len,band_frac,radius", [(600, 0.11, 33), (2000, 0.05, 50), (2100, 0.03, 31), (1000, 0.254, 127),
                                                       (900, 0.29, 130), (1100, 0.455, 250)])
def test_wide_band_modification_table_matches_oracle(lib, tmpl_len, band_frac, radius):
    """ReadType::band_width (definitions/src/lib.rs:173-175,201-210): CLR / None = ceil(0.05 L) -> radius 50 at 2 kbp; an ONT
    chunk of 2,100 bp -> radius 31.  Bit for bit the table of the oracle (and of phmm_kernel, where both take the read)."""
    import test_gpu_parity as T
    b, cfg, p = helpers.small_batch(config="ont_noisy", n_chunks=1, tmpl_len=tmpl_len, reads_per_hap=3, first=41)
    p.band_frac = band_frac
    L = len(b.template(0))
    assert int(np.ceil(L * band_frac)) // 2 == radius or abs(int(np.ceil(L * band_frac)) // 2 - radius) <= 1
    reads = list(b.chunk_reads(0))
    tab, lk = api.modification_table(p, b.template(0), [b.read(r) for r in reads], [b.read_ops(r) for r in reads],
                                     [b.strand[r] for r in reads])
    otab, olk = T.oracle_table(p, b, 0)
    assert np.array_equal(helpers.bits(lk), helpers.bits(olk))
    assert np.array_equal(helpers.bits(tab), helpers.bits(otab))
    assert (otab > -1e299).sum() > 10 * L


def test_wide_and_narrow_bands_share_a_batch(lib):
    """the radius follows each chunk's own length (mod.rs:96): 1,900 bp -> 28 (phmm_kernel), 2,100 bp -> 31
    (phmm_wide_kernel); the full path on a batch that holds both, and on a CLR-band batch"""
    piles = []
    for cid, L in [(70, 1900), (71, 2100), (72, 2150)]:
        cfg = dict(synth.CONFIGS["ont_diploid"])
        cfg.update(tmpl_len=L, reads_per_hap=6)
        piles.append(synth.make_pileup(cid, cfg, min_variants=1))
    b = jb.pack(piles)
    p = jb.default_params(haploid_coverage=6.0, band_frac=0.03)
    radii = [int(np.ceil(int(t) * 0.03)) // 2 for t in b.chunks["tmpl_len"]]
    assert min(radii) <= 30 < max(radii)
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b)
    assert_full_parity(dev, ora, b)
    b, cfg, p = helpers.small_batch(config="ont_noisy", n_chunks=2, tmpl_len=1200, reads_per_hap=6, first=88)
    p.band_frac = 0.05                                   # CLR / None (definitions/src/lib.rs:173-175): radius 30 at 1.2 kbp
    b2, cfg2, _ = helpers.small_batch(config="ont_noisy", n_chunks=2, tmpl_len=1500, reads_per_hap=6, first=90)
    dev = api.cluster_chunks(p, b2)                      # radius 37
    ora = O.cluster_chunks(helpers.oracle_params(p), b2)
    assert_full_parity(dev, ora, b2)


def test_band_wider_than_the_fallback_kernel_is_reported(lib):
    b, cfg, p = helpers.small_batch(n_chunks=2, tmpl_len=1000, reads_per_hap=3)
    p.band_frac = 0.52                                   # radius 260 > 255 (JTK_WIDE_MAX_RADIUS)
    out = api.cluster_chunks(p, b, raise_on_chunk_failure=False)
    assert out["rc"] == -6 and (out["result"]["status"] == -3).all()


# ---- the stage's model refit (model_tune.rs:96-156): polish + Baum-Welch rounds on the device

@pytest.mark.parametrize("config,rounds", [("ont_noisy", 3), ("ont_diploid", 2)])
def test_model_refit_matches_oracle(lib, config, rounds):
    """jtk_lc_fit_model against oracle/model_fit.c: every parameter of both strands' models bit for bit, after rounds of
    [polish with (band / 2, N, 0), one Baum-Welch step with the largest band]; two pile-ups of different length, so the
    fit's radius differs from one pile-up's polishing radius"""
    piles = []
    for cid, L in [(300, 500), (301, 640)]:
        cfg = dict(synth.CONFIGS[config])
        cfg.update(tmpl_len=L, reads_per_hap=6)
        piles.append(synth.make_pileup(cid, cfg))
    b = jb.pack(piles)
    p = jb.default_params(haploid_coverage=6.0, band_frac=0.03)
    p.reverse.mat_mat, p.reverse.mat_del = 0.96, 0.02          # the two strands start from different models
    rc, of, orv = O.fit_model(helpers.oracle_params(p), b, rounds=rounds)
    assert rc == 0
    df, dr = api.fit_model(p, b, rounds=rounds)
    assert bytes(df) == bytes(of) and bytes(dr) == bytes(orv)
    assert abs(df.mat_mat + df.mat_ins + df.mat_del - 1.0) < 1e-12
    assert bytes(df) != bytes(p.forward)


def test_multi_device_call_matches_single_device(lib):
    """jtk_lc_cluster_chunks_multi: LPT shares of the chunks per listed device, gathered, run, scattered back.  One GPU here,
    listed twice and three times: the partition, the gather / scatter and the stitching of the variable-length outputs are
    what is tested."""
    b, cfg, p = helpers.small_batch(n_chunks=7, tmpl_len=300, reads_per_hap=6)
    one = api.cluster_chunks(p, b)
    for devices in ([0, 0], [0, 0, 0], [0]):
        many = api.cluster_chunks(p, b, devices=devices)
        for k in ("label", "log_post", "result", "cons_off", "ops_out_off"):
            assert np.array_equal(one[k], many[k]), (devices, k)
        assert np.array_equal(one["cons"][:int(one["cons_off"][-1])], many["cons"][:int(many["cons_off"][-1])])
        assert np.array_equal(one["ops_out"][:int(one["ops_out_off"][-1])], many["ops_out"][:int(many["ops_out_off"][-1])])
    # a ragged batch: the cost-ordered deal (longest processing time first) gives every device a non-contiguous share
    shapes = ((300, 4), (420, 9), (260, 6), (380, 5), (300, 11), (340, 7), (280, 4))
    rb = jb.pack([synth.make_pileup(5000 + i, dict(synth.CONFIGS["ont_diploid"], tmpl_len=L, reads_per_hap=r))
                  for i, (L, r) in enumerate(shapes)])
    one = api.cluster_chunks(p, rb)
    for devices in ([0, 0], [0, 0, 0]):
        many = api.cluster_chunks(p, rb, devices=devices)
        for k in ("label", "log_post", "result", "cons_off", "ops_out_off"):
            assert np.array_equal(one[k], many[k]), (devices, k)
        assert np.array_equal(one["cons"][:int(one["cons_off"][-1])], many["cons"][:int(many["cons_off"][-1])])
        assert np.array_equal(one["ops_out"][:int(one["ops_out_off"][-1])], many["ops_out"][:int(many["ops_out_off"][-1])])
    with pytest.raises(ffi.JtkError) as e:
        api.cluster_chunks(p, b, devices=[])
    assert e.value.status == -1
    with pytest.raises(ffi.JtkError) as e:
        api.cluster_chunks(p, b, devices=[0, 97])
    assert e.value.status == -2


def test_heterogeneous_batch_keeps_class_0_below_80_kib(lib):
    """a deep diploid pile-up (420 reads) next to a shallow 7-copy one: each fits 80 KiB of chain work area on its own, their
    combined maxima (420 reads x 21 columns x 7 clusters) do not -- the deep one moves to the second launch class, class 0
    keeps two workgroups per CU (jtk_lc_timing_t.chain_lds_bytes), and both chunks match the oracle"""
    cfg_deep = dict(synth.CONFIGS["ont_diploid"], tmpl_len=240, reads_per_hap=210)
    cfg_wide = dict(synth.CONFIGS["ont_4copy"], tmpl_len=240, reads_per_hap=6, n_haps=7, copy_num=7, divergence=2.5e-2)
    b = jb.pack([synth.make_pileup(7100, cfg_deep, min_variants=1), synth.make_pileup(7200, cfg_wide, min_variants=2)])
    p = jb.default_params(haploid_coverage=30.0, band_frac=cfg_deep["band_frac"])
    ora = O.cluster_chunks(helpers.oracle_params(p), b, n_threads=2)
    assert ora["rc"] == 0
    dev = api.cluster_chunks(p, b)
    t = api.last_timing()
    assert 0 < t["chain_lds_bytes"][0] <= 80 * 1024 and 0 < t["chain_lds_bytes"][1] <= 160 * 1024, t["chain_lds_bytes"]
    assert t["chain_lds_bytes"][0] < t["chain_lds_bytes"][1]      # the deep pile-up went to the second class
    assert np.array_equal(dev["result"]["status"], np.zeros(2, np.int32))
    assert np.array_equal(dev["label"], ora["label"])
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"]))


def test_pileup_beyond_511_reads_matches_oracle(lib):
    """a 9-copy pile-up at 60x (540 reads) next to a diploid one: the big chunk's first pass (K = 4 on 540 reads) runs in the
    chain kernel's second launch class (LDS work area above 80 KiB), the diploid chunk in the first; both bit-exact"""
    b, p = helpers.pileup_540_inputs()
    assert int(b.chunks["n_reads"].max()) == 540
    ora = O.cluster_chunks(helpers.oracle_params(p), b, n_threads=2)
    assert ora["rc"] == 0
    dev = api.cluster_chunks(p, b)
    assert np.array_equal(dev["result"]["status"], np.zeros(2, np.int32))
    assert np.array_equal(dev["label"], ora["label"])
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert np.abs(dev["log_post"] - ora["log_post"]).max() < 1e-4
    n = int(dev["cons_off"][-1])
    assert np.array_equal(dev["cons_off"], ora["cons_off"]) and bytes(dev["cons"][:n]) == bytes(ora["cons"][:n])
