"""Synthetic inputs of the BASELINE.json configs, via jtk_synth_pileup."""
import ctypes as C

import numpy as np

from . import ffi
from .batch import pack, pileup_sort

SEED0 = 20260101

# name -> generator settings (+ stage parameters that go with it)
CONFIGS = {
    # cfg 2/3: 2-haplotype region, 60x ONT (30/hap), errors 1%/1%/1% (definitions/src/lib.rs:942-951)
    "ont_diploid": dict(tmpl_len=2000, n_haps=2, reads_per_hap=30, copy_num=2, divergence=5e-4,
                        err=(0.01, 0.01, 0.01), tmpl_err=1e-3, band_frac=0.03, coverage=30.0),
    # cfg 4: 4-copy paralogs, 80x (40/copy), pairwise divergence 1e-3
    "ont_4copy": dict(tmpl_len=2000, n_haps=4, reads_per_hap=40, copy_num=4, divergence=1e-3,
                      err=(0.01, 0.01, 0.01), tmpl_err=1e-3, band_frac=0.03, coverage=40.0),
    # cfg 5: HiFi, 40x (20/hap), 0.1% total error
    "hifi_diploid": dict(tmpl_len=2000, n_haps=2, reads_per_hap=20, copy_num=2, divergence=5e-4,
                         err=(0.001 / 3, 0.001 / 3, 0.001 / 3), tmpl_err=1e-4, band_frac=0.01, coverage=20.0),
    # stress variant of cfg 2 (benchmark_clustering.rs:22-23): 5%/5%/5%
    "ont_noisy": dict(tmpl_len=2000, n_haps=2, reads_per_hap=30, copy_num=2, divergence=5e-4,
                      err=(0.05, 0.05, 0.05), tmpl_err=1e-3, band_frac=0.03, coverage=30.0),
}


def make_pileup(chunk_id, cfg, seed0=SEED0, min_variants=0, sort=True):
    L = ffi.synth_lib()
    sc = ffi.SynthCfg(seed=seed0 + chunk_id, tmpl_len=cfg["tmpl_len"], n_haps=cfg["n_haps"],
                      reads_per_hap=cfg["reads_per_hap"], min_variants=min_variants,
                      divergence=cfg["divergence"], err_sub=cfg["err"][0], err_ins=cfg["err"][1],
                      err_del=cfg["err"][2], tmpl_err=cfg["tmpl_err"])
    n = cfg["n_haps"] * cfg["reads_per_hap"]
    cap = int(cfg["tmpl_len"] * 1.5) + 256
    tmpl = np.zeros(cap, dtype=np.uint8)
    tl = C.c_uint64(0)
    reads = np.zeros(cap * n, dtype=np.uint8)
    ops = np.zeros(2 * cap * n, dtype=np.uint8)
    read_off = np.zeros(n + 1, dtype=np.uint64)
    ops_off = np.zeros(n + 1, dtype=np.uint64)
    strand = np.zeros(n, dtype=np.uint8)
    truth = np.zeros(n, dtype=np.uint32)
    rc = L.jtk_synth_pileup(C.byref(sc), ffi.u8p(tmpl), cap, C.byref(tl), ffi.u8p(reads), len(reads),
                            ffi.u64p(read_off), ffi.u8p(ops), len(ops), ffi.u64p(ops_off), ffi.u8p(strand),
                            ffi.u32p(truth))
    if rc != 0:
        raise RuntimeError("jtk_synth_pileup failed")
    tmpl = tmpl[:tl.value].copy()
    rs = [reads[int(read_off[i]):int(read_off[i + 1])].copy() for i in range(n)]
    os_ = [ops[int(ops_off[i]):int(ops_off[i + 1])].copy() for i in range(n)]
    order = pileup_sort(tmpl, rs, os_) if sort else list(range(n))
    return (chunk_id, cfg["copy_num"], tmpl, [rs[i] for i in order], [os_[i] for i in order],
            [int(strand[i]) for i in order], [int(truth[i]) for i in order])


def make_batch(config, n_chunks, first_chunk_id=0, seed0=SEED0, min_variants=0, **overrides):
    cfg = dict(CONFIGS[config] if isinstance(config, str) else config)
    cfg.update(overrides)
    return pack([make_pileup(first_chunk_id + c, cfg, seed0, min_variants) for c in range(n_chunks)]), cfg
