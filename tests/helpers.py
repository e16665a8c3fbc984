"""Shared helpers of the parity tests."""
import ctypes as C
import os

import numpy as np

import oracle_ffi as O
from jtk_amd import batch as jb
from jtk_amd import synth


def oracle_params(p):
    """jtk_amd.ffi.Params -> oracle_ffi.Params (same C layout)."""
    return O.Params.from_buffer_copy(bytes(p))


def small_batch(config="ont_diploid", n_chunks=3, tmpl_len=400, reads_per_hap=10, first=0, min_variants=1, **kw):
    cfg = dict(synth.CONFIGS[config])
    cfg.update(tmpl_len=tmpl_len, reads_per_hap=reads_per_hap)
    cfg.update(kw)
    b, cfg = synth.make_batch(cfg, n_chunks, first_chunk_id=first, min_variants=min_variants)
    params = jb.default_params(haploid_coverage=float(reads_per_hap), band_frac=cfg["band_frac"])
    return b, cfg, params


def same_partition(a, b):
    """labels equal up to a permutation of label names"""
    a, b = np.asarray(a), np.asarray(b)
    if len(a) != len(b):
        return False
    fwd, bwd = {}, {}
    for x, y in zip(a.tolist(), b.tolist()):
        if fwd.setdefault(x, y) != y or bwd.setdefault(y, x) != x:
            return False
    return True


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


# ---- the oracle's answers for the LARGE cases, as committed fixtures -------------------------------------------------------
# The oracle needs about a second per 60-read chunk and thread (minutes for a 540-read pile-up): run inside every GPU pass it was
# most of the suite's 820 s.  Its cluster_chunks / cluster_features results are therefore looked up in tests/golden/oracle/ first,
# by CONTENT: the file name is the sha256 of the oracle's own sources (oracle_sources_sha) and of everything the oracle was given
# (inputs, parameters, flags), so a fixture can only ever answer the question it was made for BY THE ORACLE THAT IS IN THE TREE, and a
# test whose inputs change -- or an oracle that changes -- simply runs the oracle live again.  tests/test_oracle_fixtures.py re-runs
# the live oracle on two fixtures a day (CPU) and compares.  The small cases have no
# fixture: one oracle-in-the-loop case per kernel family stays in every pass.  tests/golden/make_oracle_cache.py regenerates the
# directory (it runs the listed tests with the oracle standing in for the device: no GPU needed).
ORACLE_CACHE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle")
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_oracle_sha = None


def oracle_sources_sha():
    """sha256 over everything that decides what the oracle answers: oracle/*.c, oracle/jtk_oracle.h, oracle/Makefile (its flags)
    and the arithmetic headers it shares with the device (include/jtk_math.h, include/jtk_eigen.h) -- names and contents.  It is
    part of every fixture's key (round 6; before, a fixture was keyed on the oracle's INPUTS only and a later fix to the oracle
    would have left the large GPU tests green against the old oracle's answers): change the oracle and every fixture misses,
    the tests fall back to the live oracle, and tests/golden/make_oracle_cache.py regenerates the directory."""
    global _oracle_sha
    if _oracle_sha is None:
        import hashlib
        od = os.path.join(_ROOT, "oracle")
        files = sorted(os.path.join(od, f) for f in os.listdir(od) if f.endswith(".c") or f in ("jtk_oracle.h", "Makefile"))
        files += [os.path.join(_ROOT, "include", f) for f in ("jtk_math.h", "jtk_eigen.h")]
        h = hashlib.sha256()
        for f in files:
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
        _oracle_sha = h.hexdigest()
    return _oracle_sha


def _cache_key(parts):
    import hashlib
    h = hashlib.sha256()
    h.update(oracle_sources_sha().encode())
    for a in parts:
        h.update(a if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:32]


def oracle_cache_get(key):
    path = os.path.join(ORACLE_CACHE_DIR, key + ".npz")
    if not os.path.exists(path) or os.environ.get("JTK_WRITE_GOLDEN"):
        return None
    g = np.load(path)
    return {k: (int(g[k][0]) if k == "rc" else g[k]) for k in g.files}


def oracle_cache_put(key, out, fields):
    if not os.environ.get("JTK_WRITE_GOLDEN"):
        return
    os.makedirs(ORACLE_CACHE_DIR, exist_ok=True)
    np.savez_compressed(os.path.join(ORACLE_CACHE_DIR, key + ".npz"), rc=np.array([int(out.get("rc", 0))]),
                        **{k: out[k] for k in fields})
    with open(os.path.join(ORACLE_CACHE_DIR, "INDEX.txt"), "a") as fh:
        fh.write("%s %s oracle=%s\n" % (key, os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], oracle_sources_sha()[:16]))


def cached_cluster_chunks(orig):
    """oracle_ffi.cluster_chunks behind the fixture directory (installed by tests/conftest.py)"""
    def wrapper(params, batch, skip_polish=False, n_threads=0, want_record=False):
        if want_record:
            return orig(params, batch, skip_polish=skip_polish, n_threads=n_threads, want_record=True)
        key = _cache_key([b"cluster_chunks", batch.chunks, batch.tmpl_bases, batch.read_bases, batch.read_off, batch.ops, batch.ops_off,
                          batch.strand, bytes(params), bytes([1 if skip_polish else 0])])
        hit = oracle_cache_get(key)
        if hit is not None:
            return hit
        out = orig(params, batch, skip_polish=skip_polish, n_threads=n_threads)
        n, m = int(out["cons_off"][-1]), int(out["ops_out_off"][-1])
        fields = ("label", "log_post", "result", "cons", "cons_off")
        if m <= (1 << 20):   # the re-threaded ops only where they are small (a test that compares them is a small one)
            fields += ("ops_out", "ops_out_off")
        oracle_cache_put(key, dict(out, cons=out["cons"][:n], ops_out=out["ops_out"][:m]), fields)
        return out
    wrapper._orig = orig   # (tests/test_oracle_fixtures.py runs the live oracle beside the fixture)
    return wrapper


def oracle_cluster_features(po, chunks, var, vts, stride, n_reads):
    """jo_cluster_features (oracle/local_clustering.c) behind the same fixture directory"""
    key = _cache_key([b"cluster_features", chunks, var, vts, bytes(po), bytes([stride])])
    hit = oracle_cache_get(key)
    if hit is not None:
        return hit
    from jtk_amd import ffi
    lab = np.zeros(n_reads, np.uint32)
    post = np.zeros((n_reads, stride))
    res = np.zeros(len(chunks), dtype=ffi.RESULT_DT)
    rc = O.lib().jo_cluster_features(C.byref(po), len(chunks), chunks.ctypes.data, O.f64p(var), O.u32p(vts),
                                     O.u32p(lab), O.f64p(post), stride, res.ctypes.data, 0)
    out = dict(rc=rc, label=lab, log_post=post, result=res)
    oracle_cache_put(key, out, ("label", "log_post", "result"))
    return out


def random_feature_problem(rng, n, dim, k_true):
    """feature matrix shaped like search_variants output: +gain for carriers, -gain otherwise, some zeros"""
    lab = rng.integers(0, k_true, n)
    owner = rng.integers(0, k_true, dim)
    x = np.where(lab[:, None] == owner[None, :], rng.normal(4.5, 0.8, (n, dim)), rng.normal(-4.5, 0.8, (n, dim)))
    x[rng.random((n, dim)) < 0.08] = 0.0
    x[rng.random((n, dim)) < 0.03] *= -1
    vt = np.stack([rng.integers(1, 4, dim), rng.integers(0, 3, dim)], axis=1).astype(np.uint32)
    return x, vt, lab


def feature_inputs(specs, seed):
    """the feature-level problems of a list of (n_reads, dim, k_true, copy_num): jtk_lc_feature_chunk_t records, the matrices and
    variant types back to back, the posterior stride, the read count, the planted labels"""
    from jtk_amd import ffi
    rng = np.random.default_rng(seed)
    chunks = np.zeros(len(specs), dtype=ffi.FEATURE_CHUNK_DT)
    var, vts, truth = [], [], []
    voff = vtoff = rfirst = 0
    for i, (n, dim, k_true, copy_num) in enumerate(specs):
        x, vt, lab = random_feature_problem(rng, n, dim, k_true)
        chunks[i] = (1000 + 17 * i, copy_num, n, dim, 0, voff, vtoff, rfirst, n / copy_num)
        var.append(x.ravel())
        vts.append(vt.ravel())
        truth.append(lab)
        voff += n * dim
        vtoff += dim
        rfirst += n
    return chunks, np.concatenate(var), np.concatenate(vts).astype(np.uint32), max(s[3] for s in specs), rfirst, truth


# ---- inputs of the suite's slowest device calls (single chain workgroups that run for a minute or two): the tests build them
#      here so that tests/prefetch.py can start the same calls on other host threads while the rest of the suite runs
RECURSIVE_SPLIT_CASES = [(600, 10, 2e-2, 8, 9), (800, 10, 2e-2, 10, 12), (400, 33, 2e-2, 8, 8)]   # (last: 264 reads, LDS-table chain)
LARGE_PILEUP_SPECS = ([(256, 4, 2, 2), (300, 6, 3, 3), (511, 3, 2, 2), (255, 4, 3, 3), (540, 6, 4, 4), (1023, 3, 2, 2)], 19)
HUGE_PILEUP_SPECS = ([(1024, 3, 2, 2), (60, 2, 2, 2), (1100, 4, 3, 3)], 23)


def shape_sweep_inputs(only=None):
    """the ten batches of test_random_shape_sweep_matches_oracle (a fixed-seed slice of scripts/parity_sweep_full.py): random
    configuration, template length, depth and chunk ids; every fifth batch goes through clustering_recursive's split.  Yields
    (iteration, batch, params); `only` = the iterations wanted (the generator state is advanced for the others all the same)."""
    rng = np.random.default_rng(77)
    for it in range(10):
        config = str(rng.choice(["ont_diploid", "ont_diploid", "ont_noisy", "hifi_diploid", "ont_4copy"]))
        L = int(rng.integers(130, 1400))
        rph = int(rng.integers(3, 12))
        first = int(rng.integers(0, 1 << 40))
        kw = {}
        if it % 5 == 4:
            config, L, rph = "ont_4copy", int(rng.integers(400, 900)), int(rng.integers(6, 12))
            kw = dict(n_haps=int(rng.integers(6, 11)), copy_num=int(rng.integers(8, 15)), divergence=2e-2, min_variants=3)
        if only is not None and it not in only:
            continue
        b, cfg, p = small_batch(config=config, n_chunks=3, tmpl_len=L, reads_per_hap=rph, first=first, **kw)
        yield it, b, p


def recursive_split_inputs(tmpl_len, rph, div, n_haps, copy_num):
    b, cfg, p = small_batch(config="ont_4copy", n_chunks=4, tmpl_len=tmpl_len, reads_per_hap=rph,
                            n_haps=n_haps, copy_num=copy_num, divergence=div, min_variants=3)
    b.chunks["copy_num"][3] = 2
    return b, p


def full_path_1100_inputs():
    b, cfg, p = small_batch(config="ont_diploid", n_chunks=1, tmpl_len=300, reads_per_hap=550, first=91, min_variants=1)
    return b, p


def pileup_540_inputs():
    cfg_big = dict(synth.CONFIGS["ont_4copy"], tmpl_len=260, reads_per_hap=60, n_haps=9, copy_num=9, divergence=2.5e-2)
    cfg_small = dict(synth.CONFIGS["ont_diploid"], tmpl_len=260, reads_per_hap=12)
    b = jb.pack([synth.make_pileup(4200, cfg_small), synth.make_pileup(4100, cfg_big, min_variants=3)])
    return b, jb.default_params(haploid_coverage=60.0, band_frac=cfg_big["band_frac"])


def correction_problem(seed, n_chunks=6, n_reads=40, window=(3, 6), noise=1.2, flat=0.15, wrong=0.0, single=(), cluster_dt=None, node_dt=None):
    """A small DataSet as phmm_likelihood_correction.rs sees it: two haplotypes over a chain of chunks (ids 0..n_chunks-1),
    each read covering a window of consecutive chunks on either strand; every node carries ln-posteriors over its chunk's
    clusters (two, or one for the chunks listed in `single`) and the arg-max of them as its cluster.  A fraction `flat` of
    the nodes is uninformative (posterior near 0.5/0.5), which is what the correction exists to repair;
    a fraction `wrong` is confidently mislabelled (these lower the adjusted Rand index of a chunk)."""
    from jtk_amd import ffi
    node_dt = node_dt or ffi.CC_NODE_DT
    cluster_dt = cluster_dt or ffi.CC_CHUNK_DT
    rng = np.random.default_rng(seed)
    nodes, post, node_off, hap_of = [], [], [0], []
    for r in range(n_reads):
        hap = int(rng.integers(0, 2))
        w = int(rng.integers(window[0], window[1] + 1))
        w = min(w, n_chunks)
        start = int(rng.integers(0, n_chunks - w + 1))
        fwd = bool(rng.integers(0, 2))
        ids = list(range(start, start + w))
        if not fwd:
            ids.reverse()
        for cid in ids:
            if cid in single:
                p = np.array([0.0])
                cl = 0
            else:
                z = rng.normal(noise * 3.0, noise) if rng.random() >= flat else rng.normal(0.0, 0.05)
                logit = z if hap == 1 else -z
                if wrong and rng.random() < wrong * (1 + cid % 3):  # confidently on the other haplotype, chunk dependent
                    logit = -logit
                p1 = 1.0 / (1.0 + np.exp(-logit))
                p1 = min(max(p1, 1e-6), 1.0 - 1e-6)
                p = np.log(np.array([1.0 - p1, p1]))
                cl = int(np.argmax(p))
            nodes.append((cid, cl, 1 if fwd else 0, len(p), len(post)))
            post.extend(p.tolist())
        node_off.append(len(nodes))
        hap_of.append(hap)
    nodes = np.array(nodes, dtype=node_dt)
    chunks = np.zeros(n_chunks, dtype=cluster_dt)
    chunks["id"] = np.arange(n_chunks)
    chunks["cluster_num"] = [1 if c in single else 2 for c in range(n_chunks)]
    chunks["copy_num"] = 2
    chunks["score"] = rng.uniform(0.0, 30.0, n_chunks)
    return dict(read_id=np.arange(n_reads, dtype=np.uint64) * 7 + 3, node_off=np.array(node_off, dtype=np.uint64), nodes=nodes,
                posteriors=np.array(post, dtype=np.float64), chunks=chunks, hap=np.array(hap_of))
