"""Thin numpy wrappers over the C-ABI (include/jtk_lc.h).  Every compute function runs on the GPU through
libjtk_lc.so; there is no fallback path."""
import ctypes as C

import numpy as np

from . import ffi
from .ffi import check, f64p, u8p, u32p, u64p


def _outputs(batch):
    n_chunks, n_reads, stride = batch.n_chunks, batch.n_reads, batch.post_stride
    label = np.zeros(n_reads, dtype=np.uint32)
    post = np.zeros((n_reads, stride), dtype=np.float64)
    result = np.zeros(n_chunks, dtype=ffi.RESULT_DT)
    cons_cap = int(batch.chunks["tmpl_len"].sum()) * 2 + 64 * n_chunks + 64
    ops_cap = int(len(batch.ops)) * 2 + 64 * n_reads + 64
    cons = np.zeros(cons_cap, dtype=np.uint8)
    cons_off = np.zeros(n_chunks + 1, dtype=np.uint64)
    ops_out = np.zeros(ops_cap, dtype=np.uint8)
    ops_out_off = np.zeros(n_reads + 1, dtype=np.uint64)
    return dict(label=label, log_post=post, result=result, cons=cons, cons_off=cons_off, ops_out=ops_out,
                ops_out_off=ops_out_off)


def cluster_chunks(params, batch, device=0, raise_on_chunk_failure=True, devices=None, out=None):
    """jtk_lc_cluster_chunks: polish + variant search + clustering for every chunk of `batch`;
    devices=[...]: jtk_lc_cluster_chunks_multi over that list of GPUs; out = the output arrays of an earlier call on a batch of
    this shape, to be overwritten (a host that enters the stage several times keeps its buffers)."""
    L = ffi.lib()
    o = out if out is not None else _outputs(batch)
    args = (C.byref(params), batch.n_chunks, batch.chunks.ctypes.data, u8p(batch.tmpl_bases),
            u8p(batch.read_bases), u64p(batch.read_off), u8p(batch.ops), u64p(batch.ops_off),
            u8p(batch.strand), u32p(o["label"]), f64p(o["log_post"]), batch.post_stride,
            o["result"].ctypes.data, u8p(o["cons"]), u64p(o["cons_off"]), len(o["cons"]),
            u8p(o["ops_out"]), u64p(o["ops_out_off"]), len(o["ops_out"]))
    if devices is None:
        rc = L.jtk_lc_cluster_chunks(*args, device)
    else:
        dev = (C.c_int * len(devices))(*devices)
        rc = L.jtk_lc_cluster_chunks_multi(*args, dev, len(devices))
    if rc != 0 and (raise_on_chunk_failure or rc != -6):
        check(rc)
    o["rc"] = rc
    return o


def cluster_polished(params, batch, device=0, raise_on_chunk_failure=True):
    """jtk_lc_cluster_polished: the template is already the polished consensus (pseudo_mcmc::clustering)."""
    L = ffi.lib()
    o = _outputs(batch)
    rc = L.jtk_lc_cluster_polished(C.byref(params), batch.n_chunks, batch.chunks.ctypes.data,
                                   u8p(batch.tmpl_bases), u8p(batch.read_bases), u64p(batch.read_off), u8p(batch.ops),
                                   u64p(batch.ops_off), u8p(batch.strand), u32p(o["label"]), f64p(o["log_post"]),
                                   batch.post_stride, o["result"].ctypes.data, device)
    if rc != 0 and (raise_on_chunk_failure or rc != -6):
        check(rc)
    o["rc"] = rc
    return o


def polish_chunks(params, batch, radius=0, take_num=0, ignore_edge=0, device=0, raise_on_chunk_failure=True):
    """jtk_lc_polish_chunks: polish_until_converge_antidiagonal on every window of `batch` (no clustering)."""
    L = ffi.lib()
    o = _outputs(batch)
    rc = L.jtk_lc_polish_chunks(C.byref(params), batch.n_chunks, batch.chunks.ctypes.data, u8p(batch.tmpl_bases),
                                u8p(batch.read_bases), u64p(batch.read_off), u8p(batch.ops), u64p(batch.ops_off),
                                u8p(batch.strand), radius, take_num, ignore_edge, u8p(o["cons"]), u64p(o["cons_off"]),
                                len(o["cons"]), u8p(o["ops_out"]), u64p(o["ops_out_off"]), len(o["ops_out"]),
                                o["result"].ctypes.data, device)
    if rc != 0 and (raise_on_chunk_failure or rc != -6):
        check(rc)
    o["rc"] = rc
    return o


def modification_table(params, tmpl, reads, ops, strands, device=0):
    """jtk_lc_modification_table for one pile-up -> (table [n, 14*(L+1)] minus lk, lk [n])."""
    L = ffi.lib()
    n = len(reads)
    rb = np.concatenate(reads).astype(np.uint8) if n else np.zeros(0, np.uint8)
    ob = np.concatenate(ops).astype(np.uint8) if n else np.zeros(0, np.uint8)
    ro = np.zeros(n + 1, dtype=np.uint64)
    oo = np.zeros(n + 1, dtype=np.uint64)
    ro[1:] = np.cumsum([len(r) for r in reads])
    oo[1:] = np.cumsum([len(o) for o in ops])
    st = np.array([1 if s else 0 for s in strands], dtype=np.uint8)
    tmpl = np.ascontiguousarray(tmpl, dtype=np.uint8)
    table = np.zeros((n, ffi.NUM_ROW * (len(tmpl) + 1)), dtype=np.float64)
    lk = np.zeros(n, dtype=np.float64)
    check(L.jtk_lc_modification_table(C.byref(params), u8p(tmpl), len(tmpl), n, u8p(rb), u64p(ro), u8p(ob), u64p(oo),
                                      u8p(st), f64p(table), f64p(lk), device))
    return table, lk


def estimate_gains(hmm_forward, hmm_reverse, seed=309423, seq_len=100, band=10, homop_len=3, device=0):
    """jtk_lc_estimate_gains; the defaults are estimate_gain_default's (likelihood_gains.rs:186-192)."""
    out = ffi.Gains()
    check(ffi.lib().jtk_lc_estimate_gains(C.byref(hmm_forward), C.byref(hmm_reverse), seed, seq_len, band, homop_len,
                                          C.byref(out), device))
    return out


def estimate_minimum_gain(hmm_forward, hmm_reverse, seed=23908, sample_num=1000, seq_num=500, seq_len=100, band=25, device=0):
    """jtk_lc_estimate_minimum_gain; the defaults are the reference's constants (likelihood_gains.rs:7-11)."""
    out = C.c_double(0.0)
    check(ffi.lib().jtk_lc_estimate_minimum_gain(C.byref(hmm_forward), C.byref(hmm_reverse), seed, sample_num, seq_num, seq_len,
                                                 band, C.byref(out), device))
    return out.value


def fit_model(params, batch, rounds=10, device=0):
    """jtk_lc_fit_model: the model refit of the stage preamble (model_tune.rs:119-152) on the training pile-ups `batch`;
    returns (forward, reverse)."""
    f, r = ffi.Hmm(), ffi.Hmm()
    check(ffi.lib().jtk_lc_fit_model(C.byref(params), batch.n_chunks, batch.chunks.ctypes.data, u8p(batch.tmpl_bases),
                                     u8p(batch.read_bases), u64p(batch.read_off), u8p(batch.ops), u64p(batch.ops_off),
                                     u8p(batch.strand), rounds, C.byref(f), C.byref(r), device))
    return f, r


def correct_clustering(read_id, node_off, nodes, posteriors, chunks, selection, haploid_coverage, min_gain, device=0):
    """jtk_lc_correct_clustering: AlignmentCorrection::correct_clustering_selected (phmm_likelihood_correction.rs:32-97).
    `nodes` (ffi.CC_NODE_DT) are the reads' nodes flattened by `node_off`; `chunks` (ffi.CC_CHUNK_DT) is updated in place
    (cluster_num).  Returns (cluster, touched) per node."""
    nodes = np.ascontiguousarray(nodes, dtype=ffi.CC_NODE_DT)
    posteriors = np.ascontiguousarray(posteriors, dtype=np.float64)
    read_id = np.ascontiguousarray(read_id, dtype=np.uint64)
    node_off = np.ascontiguousarray(node_off, dtype=np.uint64)
    selection = np.ascontiguousarray(selection, dtype=np.uint64)
    if chunks.dtype != ffi.CC_CHUNK_DT or not chunks.flags.c_contiguous:
        raise ValueError("chunks must be a contiguous array of ffi.CC_CHUNK_DT (it is updated in place)")
    cluster = np.zeros(len(nodes), dtype=np.uint64)
    touched = np.zeros(len(nodes), dtype=np.uint8)
    check(ffi.lib().jtk_lc_correct_clustering(len(read_id), u64p(read_id), u64p(node_off), nodes.ctypes.data, f64p(posteriors),
                                              len(chunks), chunks.ctypes.data, len(selection), u64p(selection),
                                              float(haploid_coverage), float(min_gain), u64p(cluster), u8p(touched), device))
    return cluster, touched


def cluster_features(params, feature_chunks, variants, variant_type, post_stride, device=0,
                     raise_on_chunk_failure=True):
    """jtk_lc_cluster_features: cluster_filtered_variants + posterior on caller-supplied feature matrices."""
    L = ffi.lib()
    n_reads = int(feature_chunks["n_reads"].sum())
    label = np.zeros(n_reads, dtype=np.uint32)
    post = np.zeros((n_reads, post_stride), dtype=np.float64)
    result = np.zeros(len(feature_chunks), dtype=ffi.RESULT_DT)
    variants = np.ascontiguousarray(variants, dtype=np.float64)
    variant_type = np.ascontiguousarray(variant_type, dtype=np.uint32)
    rc = L.jtk_lc_cluster_features(C.byref(params), len(feature_chunks), feature_chunks.ctypes.data, f64p(variants),
                                   u32p(variant_type), u32p(label), f64p(post), post_stride, result.ctypes.data, device)
    if rc != 0 and (raise_on_chunk_failure or rc != -6):
        check(rc)
    return dict(rc=rc, label=label, log_post=post, result=result)


def trim_cache(device=0):
    """jtk_lc_trim_cache: hand the pooled device blocks of finished calls back to the driver."""
    check(ffi.lib().jtk_lc_trim_cache(device))


def last_timing():
    t = ffi.Timing()
    check(ffi.lib().jtk_lc_last_timing(C.byref(t)))
    return dict(total_ms=t.total_ms, h2d_ms=t.h2d_ms, d2h_ms=t.d2h_ms,
                kernel_ms={n: t.kernel_ms[i] for i, n in enumerate(ffi.KERNEL_NAMES)},
                kernel_launches={n: int(t.kernel_launches[i]) for i, n in enumerate(ffi.KERNEL_NAMES)},
                chain_lds_bytes=[int(t.chain_lds_bytes[0]), int(t.chain_lds_bytes[1])])


class Session:
    """Resident-batch session: inputs uploaded once, `run()` = one pass of the hot path on the device."""

    def __init__(self, params, batch, device=0):
        self._lib = ffi.lib()
        self._h = C.c_void_p()
        self.batch = batch
        self.params = params
        check(self._lib.jtk_lc_session_create(C.byref(params), batch.n_chunks, batch.chunks.ctypes.data,
                                              u8p(batch.tmpl_bases), u8p(batch.read_bases), u64p(batch.read_off),
                                              u8p(batch.ops), u64p(batch.ops_off), u8p(batch.strand),
                                              batch.post_stride, device, C.byref(self._h)))

    def run(self, skip_polish=False):
        check(self._lib.jtk_lc_session_run(self._h, int(skip_polish)))

    def fetch(self, raise_on_chunk_failure=True, out=None):
        """every output of the stage call: labels, log-posteriors, per-chunk records, consensus and re-threaded ops (out = the
        arrays of an earlier fetch, overwritten)"""
        o = out if out is not None else _outputs(self.batch)
        rc = self._lib.jtk_lc_session_fetch(self._h, u32p(o["label"]), f64p(o["log_post"]), o["result"].ctypes.data,
                                            u8p(o["cons"]), u64p(o["cons_off"]), len(o["cons"]), u8p(o["ops_out"]),
                                            u64p(o["ops_out_off"]), len(o["ops_out"]))
        if rc != 0 and (raise_on_chunk_failure or rc != -6):
            check(rc)
        o["rc"] = rc
        return o

    def fetch_results(self, raise_on_chunk_failure=True):
        """labels, log-posteriors and the per-chunk records only (no consensus / ops): the payload of the label gather"""
        b = self.batch
        label = np.zeros(b.n_reads, dtype=np.uint32)
        post = np.zeros((b.n_reads, b.post_stride), dtype=np.float64)
        result = np.zeros(b.n_chunks, dtype=ffi.RESULT_DT)
        rc = self._lib.jtk_lc_session_fetch(self._h, u32p(label), f64p(post), result.ctypes.data, None, None, 0, None,
                                            None, 0)
        if rc != 0 and (raise_on_chunk_failure or rc != -6):
            check(rc)
        return dict(rc=rc, label=label, log_post=post, result=result)

    def trace(self, chunk):
        """the reference's trace! rows of one chunk's clustering (TOTAL / CAND / PICK / DUMP / RANGE / LK / COUNTS;
        pseudo_mcmc.rs:122-127,236,250-262,467-472,539) after run(): a list of rows"""
        need = C.c_size_t(0)
        buf = C.create_string_buffer(1 << 16)
        rc = self._lib.jtk_lc_session_trace(self._h, int(chunk), buf, len(buf), C.byref(need))
        if rc != 0 and need.value > len(buf):
            buf = C.create_string_buffer(need.value)
            rc = self._lib.jtk_lc_session_trace(self._h, int(chunk), buf, len(buf), C.byref(need))
        check(rc)
        return buf.raw[:need.value].decode().splitlines()

    def chain_profile(self):
        """jtk_lc_debug_chain_profile (include/jtk_lc_debug.h): per chunk, the cycles of its chain and its events"""
        cyc = np.zeros(self.batch.n_chunks, dtype=np.uint64)
        ev = np.zeros(self.batch.n_chunks, dtype=np.uint32)
        f = self._lib.jtk_lc_debug_chain_profile
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        check(f(self._h, u64p(cyc), u32p(ev)))
        return cyc, ev

    def close(self):
        if self._h:
            self._lib.jtk_lc_session_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def normalize_pileup(label, log_post, cluster_num):
    """normalize_local_clustering for one pile-up (normalize.rs:21-50), in place."""
    label = np.ascontiguousarray(label, dtype=np.uint32)
    log_post = np.ascontiguousarray(log_post, dtype=np.float64)
    check(ffi.lib().jtk_lc_normalize_pileup(len(label), int(cluster_num), u32p(label), f64p(log_post),
                                            log_post.shape[1]))
    return label, log_post


def record_rows(chunk_ids, n_reads, tmpl_len, read_bases_per_chunk, copy_num, result, cons_len, timing):
    """The reference's per-chunk `RECORD` lines (local_clustering/mod.rs:121: chunk id, elapsed ms, polish ms, consensus
    length, score, coverage).  The device runs the chunks of a call together, so a chunk's time is its SHARE of the call's
    kernel time per family: pair-HMM and polishing by band cells x passes, the chain by proposals x candidate k."""
    n = np.asarray(n_reads, dtype=np.float64)
    passes = np.minimum(result["polish_rounds"].astype(np.float64) + 1.0, 21.0)
    w_dp = passes * (n * np.asarray(tmpl_len, dtype=np.float64) + np.asarray(read_bases_per_chunk, dtype=np.float64))
    k_tried = np.maximum(1, np.minimum(np.asarray(copy_num, dtype=np.int64), 1 + 2 * result["n_variants"].astype(np.int64)) - 1)
    w_mc = n * k_tried * (result["n_variants"] > 0)
    km = timing["kernel_ms"]
    polish_ms = (km["phmm"] + km["polish"]) * w_dp / max(float(w_dp.sum()), 1.0)
    elapsed = polish_ms + km["filter"] / max(1, len(n)) + km["mcmc"] * w_mc / max(float(w_mc.sum()), 1.0)
    return ["RECORD\t%d\t%.3f\t%.3f\t%d\t%.3f\t%d" % (int(chunk_ids[c]), elapsed[c], polish_ms[c], int(cons_len[c]),
                                                      float(result["score"][c]), int(n[c])) for c in range(len(n))]
