"""ctypes binding of the CPU oracle (oracle/_build/libjtk_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package jtk_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libjtk_oracle.so")

NUM_ROW = 14
GAINS_MAX_HOMOP = 8


class Hmm(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("mat_mat", "mat_ins", "mat_del", "ins_mat", "ins_ins", "ins_del",
                                           "del_mat", "del_ins", "del_del")] + [
        ("mat_emit", C.c_double * 16), ("ins_emit", C.c_double * 20)]


class GainProfile(C.Structure):
    _fields_ = [("gain", C.c_double), ("prob", C.c_double)]


class Gains(C.Structure):
    _fields_ = [("max_homopolymer_len", C.c_uint32), ("reserved", C.c_uint32),
                ("subst", GainProfile * GAINS_MAX_HOMOP), ("deletions", GainProfile * GAINS_MAX_HOMOP),
                ("insertions", GainProfile * GAINS_MAX_HOMOP)]


class Params(C.Structure):
    _fields_ = [("forward", Hmm), ("reverse", Hmm), ("gains", Gains), ("haploid_coverage", C.c_double),
                ("band_frac", C.c_double)]


class Chunk(C.Structure):
    _fields_ = [("chunk_id", C.c_uint64), ("copy_num", C.c_uint32), ("n_reads", C.c_uint32),
                ("tmpl_off", C.c_uint64), ("tmpl_len", C.c_uint64), ("read_first", C.c_uint64)]


class Result(C.Structure):
    _fields_ = [("score", C.c_double), ("cluster_num", C.c_uint32), ("status", C.c_int32),
                ("polish_rounds", C.c_uint32), ("n_variants", C.c_uint32)]


class FeatureChunk(C.Structure):
    _fields_ = [("chunk_id", C.c_uint64), ("copy_num", C.c_uint32), ("n_reads", C.c_uint32),
                ("dim", C.c_uint32), ("reserved", C.c_uint32), ("var_off", C.c_uint64), ("vt_off", C.c_uint64),
                ("read_first", C.c_uint64), ("local_coverage", C.c_double)]


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4), ("draws", C.c_uint64), ("kind", C.c_uint64)]


class ClusterConfig(C.Structure):
    _fields_ = [("band_width", C.c_size_t), ("gains", C.POINTER(Gains)), ("coverage", C.c_double),
                ("copy_num", C.c_size_t), ("local_coverage", C.c_double)]


CHUNK_DT = np.dtype([("chunk_id", "<u8"), ("copy_num", "<u4"), ("n_reads", "<u4"), ("tmpl_off", "<u8"),
                     ("tmpl_len", "<u8"), ("read_first", "<u8")])
RESULT_DT = np.dtype([("score", "<f8"), ("cluster_num", "<u4"), ("status", "<i4"), ("polish_rounds", "<u4"),
                      ("n_variants", "<u4")])
FEATURE_CHUNK_DT = np.dtype([("chunk_id", "<u8"), ("copy_num", "<u4"), ("n_reads", "<u4"), ("dim", "<u4"),
                             ("reserved", "<u4"), ("var_off", "<u8"), ("vt_off", "<u8"), ("read_first", "<u8"),
                             ("local_coverage", "<f8")])
assert CHUNK_DT.itemsize == C.sizeof(Chunk) and RESULT_DT.itemsize == C.sizeof(Result)
assert FEATURE_CHUNK_DT.itemsize == C.sizeof(FeatureChunk)


def default_hmm():
    """HMMParam::default() (definitions/src/lib.rs:128-147)."""
    h = Hmm()
    for n in ("mat_mat", "ins_mat", "del_mat"):
        setattr(h, n, 0.97)
    for n in ("mat_ins", "mat_del", "ins_ins", "ins_del", "del_ins", "del_del"):
        setattr(h, n, 0.01)
    for r in range(4):
        for q in range(4):
            h.mat_emit[4 * r + q] = 0.97 if r == q else 0.01
    for i in range(20):
        h.ins_emit[i] = 0.25
    return h


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    srcs += [os.path.join(ROOT, "include", f) for f in ("jtk_lc.h", "jtk_math.h")]
    if force or not os.path.exists(ORACLE_SO) or any(
            os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


_lib = None


def u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def f64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def szp(a):
    return a.ctypes.data_as(C.POINTER(C.c_size_t))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    d, sz, u64, u32, p = C.c_double, C.c_size_t, C.c_uint64, C.c_uint32, C.c_void_p
    PD, PU8, PSZ = C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("jo_splitmix64_next", u64, C.POINTER(u64))
    sig("jo_rng_seed_from_u64", None, C.POINTER(Rng), u64)
    sig("jo_rng_next_u64", u64, C.POINTER(Rng))
    sig("jo_rng_next_u32", u32, C.POINTER(Rng))
    sig("jo_gen_range_usize", u64, C.POINTER(Rng), u64)
    sig("jo_gen_range_u32", u32, C.POINTER(Rng), u32)
    sig("jo_gen_index", u64, C.POINTER(Rng), u64)
    sig("jo_gen_bool", C.c_int, C.POINTER(Rng), d)
    sig("jo_choose_other", u64, C.POINTER(Rng), u64, u64)
    sig("jo_choose_weighted", C.c_int64, C.POINTER(Rng), PD, sz)
    sig("jo_logsumexp", d, PD, sz)
    sig("jo_rand_index", d, PSZ, PSZ, sz)
    sig("jo_kmeans", C.c_int, PD, sz, sz, sz, C.POINTER(Rng), PD, PSZ)
    sig("jo_pileup_sort_key", u64, PU8, sz, PU8, sz, PU8, sz)
    sig("jo_gains_expected", d, C.POINTER(Gains), sz, C.c_int)
    sig("jo_pvalues", None, d, sz, PD)
    sig("jo_estimate_gain", None, C.POINTER(Hmm), C.POINTER(Hmm), u64, sz, sz, sz, C.POINTER(Gains))
    sig("jo_estimate_gain_default", None, C.POINTER(Hmm), C.POINTER(Hmm), C.POINTER(Gains))
    sig("jo_homopolymer_length", None, PU8, sz, PSZ)
    sig("jo_cosine_similarity", d, PD, sz, sz, sz, sz)
    sig("jo_sokal_michener", d, PD, sz, sz, sz, sz)
    sig("jo_poisson_lk", d, sz, d)
    sig("jo_max_poisson_lk", d, sz, d, sz, sz)
    sig("jo_mcmc_with_filter", d, PD, sz, sz, PSZ, sz, d, C.POINTER(Rng))
    sig("jo_cluster_filtered_variants_exact", d, PD, sz, sz, sz, PSZ, PD)
    sig("jo_reorder_f64", None, PD, C.POINTER(u64), sz)
    sig("jo_reorder_i64", None, C.POINTER(C.c_int64), C.POINTER(u64), sz)
    sig("jo_normalize_pileup", None, sz, sz, C.POINTER(u64), PD, sz)
    sig("jo_band_centers", C.c_int, PU8, sz, sz, sz, C.POINTER(u32))
    sig("jo_phmm_likelihood", d, C.POINTER(Hmm), PU8, sz, PU8, sz, PU8, sz, sz)
    sig("jo_phmm_modification_table", d, C.POINTER(Hmm), PU8, sz, PU8, sz, PU8, sz, sz, PD)
    sig("jo_edit_ops", sz, PU8, sz, PU8, sz, PU8)
    sig("jo_phmm_likelihood_bootstrap", d, C.POINTER(Hmm), PU8, sz, PU8, sz, sz)
    sig("jo_generate_seq", None, C.POINTER(Rng), sz, PU8)
    sig("jo_phmm_gen", sz, C.POINTER(Hmm), PU8, sz, C.POINTER(Rng), PU8, sz)
    sig("jo_cluster_chunks", C.c_int, C.POINTER(Params), sz, p, PU8, PU8, C.POINTER(u64), PU8,
        C.POINTER(u64), PU8, C.c_int, C.POINTER(u32), PD, u32, p, PU8, C.POINTER(u64), u64, PU8,
        C.POINTER(u64), u64, C.c_int, PD)
    sig("jo_phmm_counts", d, C.POINTER(Hmm), PU8, sz, PU8, sz, PU8, sz, sz, PD)
    sig("jo_fit_mstep", None, C.POINTER(Hmm), PD, C.POINTER(Hmm))
    sig("jo_fit_model", C.c_int, C.POINTER(Params), sz, p, PU8, PU8, C.POINTER(u64), PU8, C.POINTER(u64), PU8, u32,
        C.POINTER(Hmm), C.POINTER(Hmm))
    sig("jo_estimate_minimum_gain", d, C.POINTER(Hmm), C.POINTER(Hmm), u64, sz, sz, sz, sz, C.c_int)
    sig("jo_correct_clustering", C.c_int, sz, C.POINTER(u64), C.POINTER(u64), p, PD, sz, p, sz, C.POINTER(u64), d, d,
        C.POINTER(u64), PU8, PD, PD)
    sig("jo_rng128pp_seed_from_u64", None, C.POINTER(Rng), u64)
    sig("jo_polish_chunks", C.c_int, C.POINTER(Params), sz, p, PU8, PU8, C.POINTER(u64), PU8, C.POINTER(u64), PU8, u32, u32,
        u32, PU8, C.POINTER(u64), PU8, C.POINTER(u64), p, C.c_int)
    sig("jo_cluster_features", C.c_int, C.POINTER(Params), sz, p, PD, C.POINTER(u32), C.POINTER(u32), PD,
        u32, p, C.c_int)
    sig("jo_modification_table", C.c_int, C.POINTER(Params), PU8, u64, u32, PU8, C.POINTER(u64), PU8,
        C.POINTER(u64), PU8, PD, PD)
    sig("jo_exp", d, d)
    sig("jo_log", d, d)
    _lib = L
    return L


def seq(s):
    """ASCII bytes -> uint8 array."""
    if isinstance(s, str):
        s = s.encode()
    return np.frombuffer(bytes(s), dtype=np.uint8).copy()


def edit_ops(tmpl, read):
    L = lib()
    ops = np.zeros(len(tmpl) + len(read) + 1, dtype=np.uint8)
    k = L.jo_edit_ops(u8p(tmpl), len(tmpl), u8p(read), len(read), u8p(ops))
    return ops[:k].copy()


def modification_table(hmm, tmpl, read, ops, radius):
    L = lib()
    tab = np.zeros(NUM_ROW * (len(tmpl) + 1), dtype=np.float64)
    lk = L.jo_phmm_modification_table(C.byref(hmm), u8p(tmpl), len(tmpl), u8p(read), len(read), u8p(ops),
                                      len(ops), radius, f64p(tab))
    return tab, lk


def likelihood(hmm, tmpl, read, ops, radius):
    return lib().jo_phmm_likelihood(C.byref(hmm), u8p(tmpl), len(tmpl), u8p(read), len(read), u8p(ops),
                                    len(ops), radius)


def cluster_chunks(params, batch, skip_polish=False, n_threads=0, want_record=False):
    """batch: jtk_amd.batch.Batch-like object with flat numpy arrays (see jtk_amd/batch.py)."""
    return _cluster_chunks_live(params, batch, skip_polish, n_threads, want_record)


def _cluster_chunks_live(params, batch, skip_polish=False, n_threads=0, want_record=False):
    """(the live oracle: tests/conftest.py puts the fixture directory in front of cluster_chunks, never of this one)"""
    L = lib()
    nchunks = len(batch.chunks)
    nreads = len(batch.strand)
    stride = batch.post_stride
    label = np.zeros(nreads, dtype=np.uint32)
    post = np.zeros((nreads, stride), dtype=np.float64)
    result = np.zeros(nchunks, dtype=RESULT_DT)
    cons_cap = int(batch.chunks["tmpl_len"].sum()) * 2 + 64 * nchunks
    ops_cap = int(len(batch.ops)) * 2 + 64 * nreads
    cons = np.zeros(cons_cap, dtype=np.uint8)
    cons_off = np.zeros(nchunks + 1, dtype=np.uint64)
    ops_out = np.zeros(ops_cap, dtype=np.uint8)
    ops_out_off = np.zeros(nreads + 1, dtype=np.uint64)
    rec = np.zeros((nchunks, 2), dtype=np.float64)
    rc = L.jo_cluster_chunks(C.byref(params), nchunks, batch.chunks.ctypes.data, u8p(batch.tmpl_bases),
                             u8p(batch.read_bases), u64p(batch.read_off), u8p(batch.ops), u64p(batch.ops_off),
                             u8p(batch.strand), int(skip_polish), u32p(label), f64p(post), stride,
                             result.ctypes.data, u8p(cons), u64p(cons_off), cons_cap, u8p(ops_out),
                             u64p(ops_out_off), ops_cap, n_threads, f64p(rec))
    out = dict(rc=rc, label=label, log_post=post, result=result, cons=cons, cons_off=cons_off,
               ops_out=ops_out, ops_out_off=ops_out_off)
    if want_record:
        out["record_ms"] = rec
    return out


class Trace(C.Structure):
    _fields_ = [("text", C.c_void_p), ("cap", C.c_size_t), ("len", C.c_size_t)]


def trace_chunk(params, batch, chunk, skip_polish=False):
    """the reference's trace! rows of one chunk's clustering as the oracle logs them (oracle/pseudo_mcmc.c: TOTAL / CAND / PICK /
    DUMP / RANGE / LK / COUNTS): the chunk alone through jo_cluster_chunks on one thread while the sink is set; a list of rows"""
    L = lib()
    L.jo_trace_set.restype = None
    L.jo_trace_set.argtypes = [C.POINTER(Trace)]
    sub = batch.subset([int(chunk)])
    buf = C.create_string_buffer(1 << 20)
    t = Trace(C.cast(buf, C.c_void_p), len(buf), 0)
    # n_threads = 1 keeps the chunk on this thread (the sink is thread-local); jo_cluster_chunks sets OpenMP's thread count for the
    # process, so it is put back afterwards: the oracle calls of the tests that follow run on every CPU again
    gomp = C.CDLL("libgomp.so.1")
    gomp.omp_get_max_threads.restype = C.c_int
    before = gomp.omp_get_max_threads()
    L.jo_trace_set(C.byref(t))
    try:
        out = _cluster_chunks_live(params, sub, skip_polish=skip_polish, n_threads=1)
    finally:
        L.jo_trace_set(None)
        gomp.omp_set_num_threads(before)
    assert t.len <= len(buf)
    return out, buf.raw[:t.len].decode().splitlines()


def polish_chunks(params, batch, radius=0, take_num=0, ignore_edge=0, n_threads=0):
    """jo_polish_chunks: polish_until_converge_antidiagonal on every window of `batch`."""
    L = lib()
    nchunks, nreads = len(batch.chunks), len(batch.strand)
    result = np.zeros(nchunks, dtype=RESULT_DT)
    cons = np.zeros(int(batch.chunks["tmpl_len"].sum()) * 2 + 64 * nchunks + 64, dtype=np.uint8)
    cons_off = np.zeros(nchunks + 1, dtype=np.uint64)
    ops_out = np.zeros(int(len(batch.ops)) * 2 + 64 * nreads + 64, dtype=np.uint8)
    ops_out_off = np.zeros(nreads + 1, dtype=np.uint64)
    rc = L.jo_polish_chunks(C.byref(params), nchunks, batch.chunks.ctypes.data, u8p(batch.tmpl_bases), u8p(batch.read_bases),
                            u64p(batch.read_off), u8p(batch.ops), u64p(batch.ops_off), u8p(batch.strand), radius, take_num,
                            ignore_edge, u8p(cons), u64p(cons_off), u8p(ops_out), u64p(ops_out_off), result.ctypes.data,
                            n_threads)
    return dict(rc=rc, result=result, cons=cons, cons_off=cons_off, ops_out=ops_out, ops_out_off=ops_out_off)


def fit_model(params, batch, rounds=10):
    """jo_fit_model: estimate_model_parameters_on_both_strands (model_tune.rs:119-152) on the training pile-ups of `batch`."""
    f, r = Hmm(), Hmm()
    rc = lib().jo_fit_model(C.byref(params), len(batch.chunks), batch.chunks.ctypes.data, u8p(batch.tmpl_bases),
                            u8p(batch.read_bases), u64p(batch.read_off), u8p(batch.ops), u64p(batch.ops_off), u8p(batch.strand),
                            rounds, C.byref(f), C.byref(r))
    return rc, f, r


def correct_clustering(read_id, node_off, nodes, posteriors, chunks, selection, haploid_coverage, min_gain, want_sims=0):
    """jo_correct_clustering (phmm_likelihood_correction.rs:32-97).  `chunks` is updated in place.  Returns
    (rc, cluster, touched, ari per chunk, similarity matrix of the first corrected chunk if want_sims = its read count)."""
    nodes = np.ascontiguousarray(nodes)
    posteriors = np.ascontiguousarray(posteriors, dtype=np.float64)
    read_id = np.ascontiguousarray(read_id, dtype=np.uint64)
    node_off = np.ascontiguousarray(node_off, dtype=np.uint64)
    selection = np.ascontiguousarray(selection, dtype=np.uint64)
    cluster = np.zeros(len(nodes), dtype=np.uint64)
    touched = np.zeros(len(nodes), dtype=np.uint8)
    ari = np.full(len(chunks), np.nan)
    sims = np.zeros((want_sims, want_sims)) if want_sims else None
    rc = lib().jo_correct_clustering(len(read_id), u64p(read_id), u64p(node_off), nodes.ctypes.data, f64p(posteriors), len(chunks),
                                     chunks.ctypes.data, len(selection), u64p(selection), float(haploid_coverage), float(min_gain),
                                     u64p(cluster), u8p(touched), f64p(ari), f64p(sims) if want_sims else None)
    return rc, cluster, touched, ari, sims
