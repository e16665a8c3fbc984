"""Shared helpers of the parity tests."""
import ctypes as C

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


def random_feature_problem(rng, n, dim, k_true):
    """feature matrix shaped like search_variants output: +gain for carriers, -gain otherwise, some zeros"""
    lab = rng.integers(0, k_true, n)
    owner = rng.integers(0, k_true, dim)
    x = np.where(lab[:, None] == owner[None, :], rng.normal(4.5, 0.8, (n, dim)), rng.normal(-4.5, 0.8, (n, dim)))
    x[rng.random((n, dim)) < 0.08] = 0.0
    x[rng.random((n, dim)) < 0.03] *= -1
    vt = np.stack([rng.integers(1, 4, dim), rng.integers(0, 3, dim)], axis=1).astype(np.uint32)
    return x, vt, lab
