"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.
Bar (BASELINE.json north_star): integer cluster assignments bit-exact, per-read log-posteriors within 1e-4;
this build additionally expects the f64 results to be bit-identical because both sides share the
arithmetic specification."""
import os

import numpy as np
import pytest

import helpers
import oracle_ffi as O
from jtk_amd import api, batch as jb, ffi, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north_star tolerance for log-likelihoods / log-posteriors


@pytest.fixture(scope="module")
def lib(jtk_lib):
    assert os.environ.get("JTK_DEVICE_IS_ORACLE") or jtk_lib.jtk_lc_device_ok(0) == 1, "needs a gfx950 device"
    return jtk_lib


def oracle_table(p, b, c):
    po = helpers.oracle_params(p)
    Lo = O.lib()
    reads = list(b.chunk_reads(c))
    t = b.template(c)
    n = len(reads)
    cols = ffi.NUM_ROW * (len(t) + 1)
    rb = np.concatenate([b.read(r) for r in reads])
    ob = np.concatenate([b.read_ops(r) for r in reads])
    ro = np.zeros(n + 1, np.uint64)
    oo = np.zeros(n + 1, np.uint64)
    ro[1:] = np.cumsum([len(b.read(r)) for r in reads])
    oo[1:] = np.cumsum([len(b.read_ops(r)) for r in reads])
    st = b.strand[reads[0]:reads[-1] + 1].copy()
    table = np.zeros((n, cols))
    lk = np.zeros(n)
    import ctypes as C
    Lo.jo_modification_table(C.byref(po), O.u8p(t), len(t), n, O.u8p(rb), O.u64p(ro), O.u8p(ob), O.u64p(oo),
                             O.u8p(st), O.f64p(table), O.f64p(lk))
    return table, lk


@pytest.mark.parametrize("tmpl_len,config", [(300, "ont_diploid"), (2000, "ont_diploid"), (700, "ont_noisy"),
                                             (2000, "hifi_diploid")])
def test_modification_table_matches_oracle(lib, tmpl_len, config):
    b, cfg, p = helpers.small_batch(config=config, n_chunks=1, tmpl_len=tmpl_len, reads_per_hap=4)
    reads = list(b.chunk_reads(0))
    tab, lk = api.modification_table(p, b.template(0), [b.read(r) for r in reads], [b.read_ops(r) for r in reads],
                                     [b.strand[r] for r in reads])
    otab, olk = oracle_table(p, b, 0)
    assert np.abs(lk - olk).max() < TOL
    finite = otab > -1e299
    assert np.array_equal(finite, tab > -1e299)
    assert np.abs(tab[finite] - otab[finite]).max() < TOL
    # this build's stronger property: identical bits
    assert np.array_equal(helpers.bits(lk), helpers.bits(olk))
    assert np.array_equal(helpers.bits(tab), helpers.bits(otab))


ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def indel_run_read(rng, tmpl, runs):
    """(`tmpl` as codes 0..3; the read comes back as ASCII bases) a read that follows `tmpl` with 4 % substitutions and, at the template positions of `runs`, a deletion (k > 0 template
    bases without a read base) or an insertion (k < 0: -k read bases without a template base) -> (read, ops)"""
    read, ops, i = [], [], 0
    runs = dict(runs)
    while i < len(tmpl):
        k = runs.pop(i, 0)
        if k > 0:
            k = min(k, len(tmpl) - i)
            ops += [3] * k
            i += k
            continue
        if k < 0:
            read += list(rng.integers(0, 4, -k))
            ops += [2] * (-k)
        if rng.random() < 0.04:
            read.append((int(tmpl[i]) + int(rng.integers(1, 4))) & 3)
            ops.append(1)
        else:
            read.append(int(tmpl[i]))
            ops.append(0)
        i += 1
    return ACGT[np.array(read, np.int64)], np.array(ops, np.uint8)


@pytest.mark.parametrize("tmpl_len,seed", [(1100, 1), (1531, 2), (1985, 3), (2000, 4)])
def test_modification_table_with_indel_runs_matches_oracle(lib, tmpl_len, seed):
    """phmm_kernel replays the forward steps of a group of 8 anti-diagonals from a checkpoint only where both sweeps take their
    unrolled path, and a run of four band moves (or none) inside half a group takes a sweep off it: reads with deletion and
    insertion runs of 3-15 bases at every alignment of the run against the groups, near both ends of the sweep and across
    the 64-diagonal scaling blocks, so that replayed, streamed and generic groups meet in every order"""
    rng = np.random.default_rng(seed)
    p = jb.default_params(haploid_coverage=25.0)
    tmpl = rng.integers(0, 4, tmpl_len).astype(np.uint8)
    reads, ops = [], []
    for r in range(6):
        runs, pos = {}, int(rng.integers(3, 40))
        while pos < tmpl_len - 20:
            k = int(rng.integers(3, 16))
            runs[pos] = k if rng.random() < 0.6 else -k
            pos += k + int(rng.integers(1, 90))   # sometimes two runs within one group of diagonals
        if r == 0:
            runs = {}                               # one plain read: every inner group replays
        rd, op = indel_run_read(rng, tmpl, runs)
        reads.append(rd)
        ops.append(op)
    strands = [1, 0, 1, 0, 1, 1]
    tmpl = ACGT[tmpl]
    tab, lk = api.modification_table(p, tmpl, reads, ops, strands)
    po = helpers.oracle_params(p)
    import ctypes as C
    rb, ob = np.concatenate(reads), np.concatenate(ops)
    ro = np.zeros(7, np.uint64)
    oo = np.zeros(7, np.uint64)
    ro[1:] = np.cumsum([len(x) for x in reads])
    oo[1:] = np.cumsum([len(x) for x in ops])
    otab = np.zeros_like(tab)
    olk = np.zeros(6)
    O.lib().jo_modification_table(C.byref(po), O.u8p(tmpl), tmpl_len, 6, O.u8p(rb), O.u64p(ro), O.u8p(ob), O.u64p(oo),
                                  O.u8p(np.array(strands, np.uint8)), O.f64p(otab), O.f64p(olk))
    assert np.array_equal(helpers.bits(lk), helpers.bits(olk))
    assert np.array_equal(helpers.bits(tab), helpers.bits(otab))


def test_modification_table_rejects_inconsistent_ops(lib):
    b, cfg, p = helpers.small_batch(n_chunks=1, tmpl_len=200, reads_per_hap=2)
    reads = list(b.chunk_reads(0))
    ops = [b.read_ops(r).copy() for r in reads]
    ops[1] = ops[1][:-3]
    with pytest.raises(ffi.JtkError) as e:
        api.modification_table(p, b.template(0), [b.read(r) for r in reads], ops, [1, 1, 1, 1])
    assert e.value.status == -5


@pytest.mark.parametrize("seed,seq_len,band,homop_len", [(309423, 100, 10, 3), (77, 60, 8, 2)])
def test_estimate_gains_matches_oracle(lib, seed, seq_len, band, homop_len):
    """the stage preamble estimate_gain (likelihood_gains.rs:162-192): host sampling, device bootstrap alignments and
    banded likelihoods, against the oracle's all-CPU restatement -- gains and null probabilities of all 9 profiles"""
    import ctypes as C
    p = jb.default_params(haploid_coverage=30.0)
    if seed != 309423:                     # strands with different models (update_models_on_both_strands fits two)
        p.reverse.mat_mat -= 0.01
        p.reverse.mat_del += 0.01
    dev = api.estimate_gains(p.forward, p.reverse, seed, seq_len, band, homop_len)
    po = helpers.oracle_params(p)
    ora = O.Gains()
    O.lib().jo_estimate_gain(C.byref(po.forward), C.byref(po.reverse), seed, seq_len, band, homop_len, C.byref(ora))
    assert dev.max_homopolymer_len == ora.max_homopolymer_len == homop_len
    for name in ("subst", "deletions", "insertions"):
        for h in range(homop_len):
            d, o = getattr(dev, name)[h], getattr(ora, name)[h]
            assert abs(d.gain - o.gain) < TOL and abs(d.prob - o.prob) < TOL, (name, h, d.gain, o.gain, d.prob, o.prob)
            assert d.gain == o.gain and d.prob == o.prob      # this build: identical bits
            assert d.gain > 0.0 and 1e-9 <= d.prob <= 1.0


@pytest.mark.parametrize("seed,sample_num,seq_num,seq_len,band", [(23908, 40, 60, 100, 25), (5, 7, 21, 64, 12)])
def test_estimate_minimum_gain_matches_oracle(lib, seed, sample_num, seq_num, seq_len, band):
    """estimate_minimum_gain (likelihood_gains.rs:6-39), the scale of correct_clustering's protection rule: the reference's
    seed / length / band on a reduced number of samples and reads, and an odd shape; identical bits"""
    import ctypes as C
    p = jb.default_params(haploid_coverage=30.0)
    p.reverse.mat_mat -= 0.01
    p.reverse.mat_ins += 0.01
    dev = api.estimate_minimum_gain(p.forward, p.reverse, seed, sample_num, seq_num, seq_len, band)
    po = helpers.oracle_params(p)
    ora = O.lib().jo_estimate_minimum_gain(C.byref(po.forward), C.byref(po.reverse), seed, sample_num, seq_num, seq_len, band, 8)
    assert dev == ora and dev >= 1.0
    with pytest.raises(ffi.JtkError):
        api.estimate_minimum_gain(p.forward, p.reverse, seed, 2, seq_num, seq_len, band)


def random_feature_problem(rng, n, dim, k_true, cid=0, copy_num=2):
    return helpers.random_feature_problem(rng, n, dim, k_true)


def run_features_both(p, specs, seed):
    chunks, var, vts, stride, rfirst, truth = helpers.feature_inputs(specs, seed)
    ora = helpers.oracle_cluster_features(helpers.oracle_params(p), chunks, var, vts, stride, rfirst)   # (large cases: a fixture)
    assert ora["rc"] == 0
    dev = api.cluster_features(p, chunks, var, vts, stride)
    return dev, ora, truth


def test_cluster_features_matches_oracle(lib):
    p = jb.default_params(haploid_coverage=12.0)
    specs = [(24, 3, 2, 2), (24, 6, 2, 2), (30, 1, 2, 2), (36, 5, 3, 3), (40, 6, 3, 4), (20, 4, 1, 2),
             (6, 3, 2, 2), (2, 2, 2, 2), (25, 0, 2, 2), (20, 3, 2, 1)]
    dev, ora, truth = run_features_both(p, specs, seed=1)
    assert np.array_equal(dev["label"], ora["label"])
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert np.abs(dev["log_post"] - ora["log_post"]).max() < TOL
    assert np.abs(dev["result"]["score"] - ora["result"]["score"]).max() < TOL
    assert np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"]))
    assert np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(ora["result"]["score"]))
    # rows of the posterior logsumexp to 0 (mod.rs:184-185 invariant)
    off = 0
    for (n, dim, kt, cn), k in zip(specs, dev["result"]["cluster_num"]):
        rows = dev["log_post"][off:off + n, :k]
        assert np.abs(np.log(np.exp(rows).sum(axis=1))).max() < 1e-9
        off += n
    # sanity: the clean 2-cluster problems are recovered
    assert helpers.same_partition(dev["label"][:24], truth[0]) or (dev["label"][:24] == truth[0]).mean() > 0.9


def test_cluster_polished_matches_oracle(lib):
    b, cfg, p = helpers.small_batch(n_chunks=4, tmpl_len=500, reads_per_hap=10)
    dev = api.cluster_polished(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=True)
    assert ora["rc"] == 0
    assert np.array_equal(dev["result"]["n_variants"], ora["result"]["n_variants"])
    assert np.array_equal(dev["label"], ora["label"])
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert np.abs(dev["log_post"] - ora["log_post"]).max() < TOL
    assert np.abs(dev["result"]["score"] - ora["result"]["score"]).max() < TOL


@pytest.mark.parametrize("config,tmpl_len,rph,nch", [("ont_diploid", 500, 10, 4), ("ont_diploid", 2000, 30, 2),
                                                     ("ont_4copy", 600, 8, 2), ("hifi_diploid", 800, 10, 2)])
def test_cluster_chunks_matches_oracle(lib, config, tmpl_len, rph, nch):
    b, cfg, p = helpers.small_batch(config=config, n_chunks=nch, tmpl_len=tmpl_len, reads_per_hap=rph)
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b, skip_polish=False)
    assert ora["rc"] == 0
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
    for c in range(b.n_chunks):
        k = int(dev["result"][c]["cluster_num"])
        rows = dev["log_post"][list(b.chunk_reads(c))][:, :k]
        assert np.abs(np.log(np.exp(rows).sum(axis=1))).max() < 1e-4       # mod.rs:184-185


def test_edge_cases(lib):
    """empty batch, empty pile-up, copy_num 0/1, a pile-up smaller than its copy number"""
    p = jb.default_params(haploid_coverage=5.0)
    empty = jb.pack([])
    out = api.cluster_chunks(p, empty)
    assert len(out["label"]) == 0
    b, cfg, _ = helpers.small_batch(n_chunks=3, tmpl_len=300, reads_per_hap=3)
    b.chunks["copy_num"][0] = 1
    b.chunks["copy_num"][1] = 6       # n = 6 <= copy_num -> trivial (pseudo_mcmc.rs:221)
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b)
    assert np.array_equal(dev["label"], ora["label"])
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert dev["result"]["cluster_num"][0] == 1 and dev["result"]["cluster_num"][1] == 1
    assert np.abs(dev["log_post"] - ora["log_post"]).max() < TOL
    # copy_num >= 8 on a small diploid pile-up: the split branch finds one cluster and returns it (mod.rs:146-148)
    b.chunks["copy_num"][2] = 8
    dev = api.cluster_chunks(p, b)
    ora = O.cluster_chunks(helpers.oracle_params(p), b)
    assert np.array_equal(dev["label"], ora["label"])
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert np.abs(dev["log_post"] - ora["log_post"]).max() < TOL


@pytest.mark.parametrize("tmpl_len,rph,div,n_haps,copy_num", helpers.RECURSIVE_SPLIT_CASES)
def test_recursive_split_matches_oracle(lib, tmpl_len, rph, div, n_haps, copy_num):
    """copy_num >= 8: clustering_recursive's split branch (mod.rs:138-189) -- a 4-way clustering, then per group a
    consensus polish and a clustering with the group's share of the copies, all on one RNG stream per chunk;
    copy_num 12 nests a second split.  The last chunk keeps its reads but is declared diploid, so that split and
    plain chunks share a batch."""
    b, p = helpers.recursive_split_inputs(tmpl_len, rph, div, n_haps, copy_num)
    ora = O.cluster_chunks(helpers.oracle_params(p), b)
    assert ora["rc"] == 0
    assert ora["result"]["cluster_num"][:3].max() > 4, "the inputs must exercise the merge of sub-clusterings"
    dev = api.cluster_chunks(p, b)
    assert np.array_equal(dev["label"], ora["label"])
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert np.abs(dev["log_post"] - ora["log_post"]).max() < TOL
    assert np.abs(dev["result"]["score"] - ora["result"]["score"]).max() < TOL
    n = int(dev["cons_off"][-1])
    assert np.array_equal(dev["cons_off"], ora["cons_off"]) and bytes(dev["cons"][:n]) == bytes(ora["cons"][:n])
    m = int(dev["ops_out_off"][-1])
    assert np.array_equal(dev["ops_out"][:m], ora["ops_out"][:m])
    for c in range(b.n_chunks):
        k = int(dev["result"][c]["cluster_num"])
        rows = dev["log_post"][list(b.chunk_reads(c))][:, :k]
        assert np.abs(np.log(np.exp(rows).sum(axis=1))).max() < 1e-4       # mod.rs:184-185
    if rph > 30 or os.environ.get("JTK_DEVICE_IS_ORACLE"):
        return       # (the large pile-ups spend ~100 s in the generic chain: one run is enough)
    # a second run of the same session repeats the recursion from the chunk seeds
    with api.Session(p, b) as s:
        s.run()
        again = s.fetch()
    assert np.array_equal(again["label"], dev["label"]) and np.array_equal(again["log_post"], dev["log_post"])


def test_concurrent_sessions_match_one_shot(lib):
    """sessions are independent (own stream, own workspaces): four of them driven from four host threads, as bench.py
    does to overlap one batch's chain tail with another's pair-HMM passes, give the one-shot results"""
    import threading
    b, cfg, p = helpers.small_batch(n_chunks=6, tmpl_len=500, reads_per_hap=10)
    one = api.cluster_chunks(p, b)
    sessions = [api.Session(p, b) for _ in range(4)]
    errors = []

    def worker(s):
        try:
            for _ in range(3):
                s.run()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(s,)) for s in sessions]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for s in sessions:
        out = s.fetch()
        for k in ("label", "log_post", "cons", "ops_out"):
            assert np.array_equal(out[k], one[k])
        s.close()


@pytest.mark.changes_env
@pytest.mark.parametrize("n_slices", [2, 3, 7])
def test_one_shot_slicing_does_not_change_results(lib, monkeypatch, n_slices):
    """jtk_lc_cluster_chunks runs a large batch as up to four slices on their own streams and host threads (>= 500
    chunks each); JTK_LC_SLICES forces that path on a small ragged batch: every output, including the stitched
    consensus / ops offsets, must equal the unsliced call"""
    b, cfg, p = helpers.small_batch(n_chunks=7, tmpl_len=350, reads_per_hap=6)
    b.chunks["copy_num"][2] = 1          # a trivial chunk inside a slice
    monkeypatch.setenv("JTK_LC_SLICES", "1")
    whole = api.cluster_chunks(p, b)
    whole_pol = api.cluster_polished(p, b)
    monkeypatch.setenv("JTK_LC_SLICES", str(n_slices))
    sliced = api.cluster_chunks(p, b)
    for k in ("label", "log_post", "cons", "cons_off", "ops_out", "ops_out_off"):
        assert np.array_equal(sliced[k], whole[k]), k
    for f in ("score", "cluster_num", "status", "polish_rounds", "n_variants"):
        assert np.array_equal(sliced["result"][f], whole["result"][f]), f
    sliced_pol = api.cluster_polished(p, b)
    assert np.array_equal(sliced_pol["label"], whole_pol["label"])
    assert np.array_equal(sliced_pol["log_post"], whole_pol["log_post"])
    t = api.last_timing()
    assert t["kernel_launches"]["mcmc"] == min(n_slices, 7)


def _copy_batch(b):
    import dataclasses
    return dataclasses.replace(b, **{f.name: getattr(b, f.name).copy() for f in dataclasses.fields(b)
                                    if isinstance(getattr(b, f.name), np.ndarray)})


@pytest.mark.changes_env
def test_device_side_validation_reports_what_the_host_loop_reported(lib, monkeypatch):
    """Round 6: reads and ops are validated and recoded by encode_reads_kernel, not by a host loop.  Same verdicts: a non-ACGT
    base in a read or a template and an op code above 3 fail the CALL with JTK_ERR_INVALID_ARG (also when the offender sits in
    the last slice of a sliced call); lower-case bases are bases."""
    b, cfg, p = helpers.small_batch(n_chunks=6, tmpl_len=260, reads_per_hap=5)
    good = api.cluster_chunks(p, b)
    low = _copy_batch(b)
    low.read_bases[:] = np.frombuffer(bytes(low.read_bases).lower(), dtype=np.uint8)
    low.tmpl_bases[:] = np.frombuffer(bytes(low.tmpl_bases).lower(), dtype=np.uint8)
    out = api.cluster_chunks(p, low)
    for k in ("label", "log_post", "cons", "cons_off", "ops_out", "ops_out_off"):
        assert np.array_equal(out[k], good[k]), k
    last_read = int(b.read_off[-2])           # first base of the batch's last read
    for slices in ("1", "3"):
        monkeypatch.setenv("JTK_LC_SLICES", slices)
        bad = _copy_batch(b)
        bad.read_bases[last_read + 7] = ord("N")
        with pytest.raises(ffi.JtkError) as e:
            api.cluster_chunks(p, bad)
        assert e.value.status == -1 and "non-ACGT base in a read" in str(e.value)
        bad = _copy_batch(b)
        bad.ops[int(b.ops_off[-2]) + 3] = 7
        with pytest.raises(ffi.JtkError) as e:
            api.cluster_chunks(p, bad)
        assert e.value.status == -1 and "bad op code" in str(e.value)
        bad = _copy_batch(b)
        bad.tmpl_bases[int(b.chunks["tmpl_off"][5]) + 11] = ord("-")
        with pytest.raises(ffi.JtkError) as e:
            api.cluster_chunks(p, bad)
        assert e.value.status == -1 and "template" in str(e.value)
    assert np.array_equal(api.cluster_chunks(p, b)["label"], good["label"])   # (the library is fine afterwards)


@pytest.mark.changes_env
def test_a_failed_chunk_inside_a_slice_leaves_the_other_outputs_in_place(lib, monkeypatch):
    """One read whose ops do not consume its template (band_prep: JTK_ERR_OPS_MISMATCH for the chunk): the call returns
    JTK_ERR_CHUNK_FAILED, the chunk has no consensus / ops (empty ranges), and every other chunk's results -- written by the
    slices straight into the caller's arrays at offsets that depend on the failed chunk's ZERO length -- equal those of the same
    call in one piece."""
    b, cfg, p = helpers.small_batch(n_chunks=7, tmpl_len=300, reads_per_hap=5)
    bad = _copy_batch(b)
    r = int(bad.chunks["read_first"][3]) + 2
    o0 = int(bad.ops_off[r])
    k = next(i for i in range(o0, int(bad.ops_off[r + 1])) if bad.ops[i] == ffi.OP_MATCH)
    bad.ops[k] = ffi.OP_INS                     # one template base is no longer consumed
    monkeypatch.setenv("JTK_LC_SLICES", "1")
    whole = api.cluster_chunks(p, bad, raise_on_chunk_failure=False)
    assert whole["rc"] == -6 and int(whole["result"]["status"][3]) == -5
    assert (np.delete(whole["result"]["status"], 3) == 0).all()
    assert whole["cons_off"][3] == whole["cons_off"][4]                      # nothing for the failed chunk
    ref = api.cluster_chunks(p, b)
    for c in (0, 1, 2, 4, 5, 6):                                             # the others are what they are without it
        a0, a1 = int(whole["cons_off"][c]), int(whole["cons_off"][c + 1])
        r0, r1 = int(ref["cons_off"][c]), int(ref["cons_off"][c + 1])
        assert bytes(whole["cons"][a0:a1]) == bytes(ref["cons"][r0:r1])
    for n_slices in ("2", "3", "7"):
        monkeypatch.setenv("JTK_LC_SLICES", n_slices)
        sliced = api.cluster_chunks(p, bad, raise_on_chunk_failure=False)
        assert sliced["rc"] == -6
        for key in ("label", "log_post", "cons_off", "ops_out_off"):
            assert np.array_equal(sliced[key], whole[key]), (n_slices, key)
        n, m = int(whole["cons_off"][-1]), int(whole["ops_out_off"][-1])
        assert np.array_equal(sliced["cons"][:n], whole["cons"][:n]) and np.array_equal(sliced["ops_out"][:m], whole["ops_out"][:m])
        assert np.array_equal(sliced["result"]["status"], whole["result"]["status"])


def test_session_is_repeatable_and_matches_one_shot(lib):
    b, cfg, p = helpers.small_batch(n_chunks=3, tmpl_len=400, reads_per_hap=8)
    one = api.cluster_chunks(p, b)
    with api.Session(p, b) as s:
        s.run()
        a = s.fetch()
        s.run()
        c = s.fetch()
    for k in ("label", "log_post", "cons", "ops_out"):
        assert np.array_equal(a[k], c[k]) and np.array_equal(a[k], one[k])
    t = api.last_timing()
    assert t["kernel_launches"]["phmm"] >= 1


def test_chain_variants_match_oracle(lib):
    """every code path of the diploid chain and its neighbours: replicated state (1, 2, 3-4 columns), the
    lane-distributed state (5-8), more columns / more reads than the table-driven chain takes (generic chain on
    the same stream), all-zero rows (size-only moves), tiny pile-ups"""
    p = jb.default_params(haploid_coverage=15.0)
    specs = [(60, 1, 2, 2), (60, 2, 2, 2), (45, 4, 2, 2), (63, 8, 2, 2), (40, 7, 2, 2), (50, 9, 2, 2),
             (64, 3, 2, 2), (12, 4, 2, 2), (5, 2, 2, 2), (33, 5, 1, 2),
             # more than 63 reads: two table registers per read / per cluster size
             (65, 1, 2, 2), (100, 6, 2, 2), (127, 2, 2, 2), (90, 4, 2, 2), (128, 3, 2, 2)]
    dev, ora, truth = run_features_both(p, specs, seed=7)
    assert np.array_equal(dev["label"], ora["label"])
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"]))
    assert np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(ora["result"]["score"]))


def test_large_pileups_match_oracle(lib):
    """more than 255 reads (7 copies x 40 reads, or the first pass of a copy_num >= 8 chunk): the generic chain with its
    per-read and per-size tables in LDS instead of registers; 256 and 1023 (10-bit read indices) are the edges of that
    mode, 540 reads is a 9-copy pile-up at 60x"""
    p = jb.default_params(haploid_coverage=40.0)
    specs, seed = helpers.LARGE_PILEUP_SPECS
    dev, ora, truth = run_features_both(p, specs, seed=seed)
    assert np.array_equal(dev["result"]["status"], np.zeros(len(specs), np.int32))
    assert np.array_equal(dev["label"], ora["label"])
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"]))
    assert np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(ora["result"]["score"]))


def test_pileups_beyond_1023_reads_match_oracle(lib):
    """clustering_on_pileup (mod.rs:86-123) takes any depth.  Beyond JTK_MAX_PILEUP = 1,023 reads (10-bit read indices of the
    table-driven chains) or a CU's LDS, a pile-up runs in mcmc_kernel_huge: work area in global memory, one proposal per
    iteration -- slow, and bit-exact: 1,024 and 1,100 reads, K = 2 and 3, next to an ordinary pile-up in the same call"""
    p = jb.default_params(haploid_coverage=40.0)
    specs, seed = helpers.HUGE_PILEUP_SPECS
    dev, ora, truth = run_features_both(p, specs, seed=seed)
    assert np.array_equal(dev["result"]["status"], np.zeros(len(specs), np.int32))
    assert np.array_equal(dev["label"], ora["label"])
    assert np.array_equal(dev["result"]["cluster_num"], ora["result"]["cluster_num"])
    assert np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"]))
    assert np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(ora["result"]["score"]))


def test_high_acceptance_chains_match_oracle(lib):
    """weak columns (|gain| ~ 0.3: what the refitted model of the real stage lets through the filter): half of the proposals are
    accepted, and with three or more columns the diploid chain then takes every proposal as one exact step (direct mode) instead
    of rebuilding its tables after every accepted move; mixed with strong columns so that it goes in and out of that mode"""
    import ctypes as C
    p = jb.default_params(haploid_coverage=25.0)
    rng = np.random.default_rng(41)
    specs = [(50, 3), (60, 4), (44, 6), (100, 3), (60, 8), (63, 5)]
    chunks = np.zeros(len(specs), dtype=ffi.FEATURE_CHUNK_DT)
    var, vts = [], []
    voff = vtoff = rfirst = 0
    for i, (n, dim) in enumerate(specs):
        x, vt, _ = random_feature_problem(rng, n, dim, 2, i, 2)
        weak = rng.random(dim) < 0.7
        x[:, weak] = rng.normal(0.0, 0.3, (n, int(weak.sum())))
        x[np.abs(x) < 2e-5] = 0.0   # (LKCount asserts on |x| == POS_THR exactly; keep clear of it)
        chunks[i] = (900 + 11 * i, 2, n, dim, 0, voff, vtoff, rfirst, n / 2)
        var.append(x.ravel())
        vts.append(vt.ravel())
        voff += n * dim
        vtoff += dim
        rfirst += n
    var = np.concatenate(var)
    vts = np.concatenate(vts).astype(np.uint32)
    dev = api.cluster_features(p, chunks, var, vts, 2)
    po = helpers.oracle_params(p)
    lab = np.zeros(rfirst, np.uint32)
    post = np.zeros((rfirst, 2))
    res = np.zeros(len(specs), dtype=ffi.RESULT_DT)
    assert O.lib().jo_cluster_features(C.byref(po), len(specs), chunks.ctypes.data, O.f64p(var), O.u32p(vts),
                                       O.u32p(lab), O.f64p(post), 2, res.ctypes.data, 0) == 0
    assert np.array_equal(dev["label"], lab)
    assert np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(post))
    assert np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(res["score"]))


def test_size_only_moves_match_oracle(lib):
    """a third of the reads carry no signal at all (all-zero rows): hundreds of thousands of accepted moves that
    change nothing but the cluster sizes, decided from the per-size tables"""
    import ctypes as C
    p = jb.default_params(haploid_coverage=20.0)
    rng = np.random.default_rng(11)
    specs = [(60, 1), (48, 2), (60, 3), (96, 1), (120, 2)]
    chunks = np.zeros(len(specs), dtype=ffi.FEATURE_CHUNK_DT)
    var, vts = [], []
    voff = vtoff = rfirst = 0
    for i, (n, dim) in enumerate(specs):
        x, vt, _ = random_feature_problem(rng, n, dim, 2, i, 2)
        x[rng.random(n) < 0.35] = 0.0
        chunks[i] = (77 + 5 * i, 2, n, dim, 0, voff, vtoff, rfirst, n / 2)
        var.append(x.ravel())
        vts.append(vt.ravel())
        voff += n * dim
        vtoff += dim
        rfirst += n
    var = np.concatenate(var)
    vts = np.concatenate(vts).astype(np.uint32)
    dev = api.cluster_features(p, chunks, var, vts, 2)
    po = helpers.oracle_params(p)
    lab = np.zeros(rfirst, np.uint32)
    post = np.zeros((rfirst, 2))
    res = np.zeros(len(specs), dtype=ffi.RESULT_DT)
    assert O.lib().jo_cluster_features(C.byref(po), len(specs), chunks.ctypes.data, O.f64p(var), O.u32p(vts),
                                       O.u32p(lab), O.f64p(post), 2, res.ctypes.data, 0) == 0
    assert np.array_equal(dev["label"], lab)
    assert np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(post))
    assert np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(res["score"]))


# ---- BASELINE.json's full sizes (60 reads x 2 kbp per chunk): properties that need no oracle run

@pytest.fixture(scope="module")
def full_size(lib):
    cfg = dict(synth.CONFIGS["ont_diploid"])
    b, cfg = synth.make_batch(cfg, 24)
    p = jb.default_params(cfg["coverage"], cfg["band_frac"])
    return b, p, api.cluster_chunks(p, b)


def test_full_size_results_do_not_depend_on_batch_order(lib, full_size):
    """chunks are independent (own RNG stream from the chunk id): any sub-batch, in any order, reproduces them"""
    b, p, out = full_size
    idx = [17, 3, 11, 0, 23]
    sub = b.subset(idx)
    o2 = api.cluster_chunks(p, sub)
    for k, c in enumerate(idx):
        ra, rb_ = list(b.chunk_reads(c)), list(sub.chunk_reads(k))
        assert np.array_equal(out["label"][ra], o2["label"][rb_])
        assert np.array_equal(helpers.bits(out["log_post"][ra]), helpers.bits(o2["log_post"][rb_]))
        assert out["result"][c]["cluster_num"] == o2["result"][k]["cluster_num"]
        assert out["result"][c]["polish_rounds"] == o2["result"][k]["polish_rounds"]
        ca = out["cons"][int(out["cons_off"][c]):int(out["cons_off"][c + 1])]
        cb = o2["cons"][int(o2["cons_off"][k]):int(o2["cons_off"][k + 1])]
        assert bytes(ca) == bytes(cb)


def test_full_size_polish_is_a_fixed_point(lib, full_size):
    """feeding the polished consensus and the re-threaded alignments back in changes nothing: no edit is found,
    and the clustering (tables -> variants -> chain) is reproduced bit for bit"""
    b, p, out = full_size
    pile = []
    for c in range(8):
        reads = list(b.chunk_reads(c))
        cons = out["cons"][int(out["cons_off"][c]):int(out["cons_off"][c + 1])]
        ops = [out["ops_out"][int(out["ops_out_off"][r]):int(out["ops_out_off"][r + 1])] for r in reads]
        pile.append((int(b.chunks[c]["chunk_id"]), int(b.chunks[c]["copy_num"]), cons, [b.read(r) for r in reads], ops,
                     [int(b.strand[r]) for r in reads], None))
    again = api.cluster_chunks(p, jb.pack(pile))
    n8 = int(b.chunks[8]["read_first"])
    assert np.all(again["result"]["polish_rounds"] <= 1)
    assert bytes(again["cons"][:int(again["cons_off"][8])]) == bytes(out["cons"][:int(out["cons_off"][8])])
    # the band radius follows the template length (mod.rs:96), so only chunks whose length did not move are
    # guaranteed the same band; for those the whole downstream path must repeat exactly
    for c in range(8):
        L0, L1 = int(b.chunks[c]["tmpl_len"]), int(out["cons_off"][c + 1] - out["cons_off"][c])
        if int(np.ceil(L0 * p.band_frac)) == int(np.ceil(L1 * p.band_frac)):
            ra = list(b.chunk_reads(c))
            assert np.array_equal(again["label"][ra], out["label"][ra])
            assert np.array_equal(helpers.bits(again["log_post"][ra]), helpers.bits(out["log_post"][ra]))
    assert n8 == len(again["label"])


def test_full_size_posteriors_are_normalised_and_labels_recover_truth(lib, full_size):
    b, p, out = full_size
    agree = []
    for c in range(b.n_chunks):
        rr = list(b.chunk_reads(c))
        k = int(out["result"][c]["cluster_num"])
        rows = out["log_post"][rr][:, :k]
        assert np.abs(np.log(np.exp(rows).sum(axis=1))).max() < 1e-4   # mod.rs:184-185
        assert out["label"][rr].max() < k
        if k == 2:
            lab, tr = out["label"][rr], b.truth[rr]
            agree.append(max((lab == tr).mean(), (lab != tr).mean()))
    assert len(agree) >= 6 and np.mean(agree) > 0.9


@pytest.mark.parametrize("seed", [20, 21, 23])
def test_size_only_moves_after_residues_match_oracle(lib, seed):
    """regression: `proposed - lk` of a size-only move uses the lk the chain CARRIES (flip-back residues move the
    column sums, not lk), not a difference of two table entries; read counts just above a power of two make
    gen_range reject ~45 % of its draws, which is when one misjudged 'draws nothing' shifts the whole stream"""
    p = jb.default_params(haploid_coverage=15.0)
    specs = [(65, 1, 2, 2), (65, 2, 2, 2), (67, 1, 2, 2), (69, 1, 2, 2), (71, 1, 2, 2), (97, 1, 2, 2), (75, 1, 2, 2),
             (85, 2, 2, 2), (61, 1, 2, 2)]
    dev, ora, truth = run_features_both(p, specs, seed=seed)
    assert np.array_equal(dev["label"], ora["label"])
    assert np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"]))
    assert np.array_equal(helpers.bits(dev["result"]["score"]), helpers.bits(ora["result"]["score"]))
