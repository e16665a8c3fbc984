"""ctypes binding of libjtk_lc.so (the C-ABI of include/jtk_lc.h) and of libjtk_synth.so (include/jtk_synth.h: the
synthetic pile-up generator of bench.py and the tests, deliberately a separate library).

There is no Python or CPU fallback: if the shared library is missing this module raises, and a compute
call on a machine without a gfx950 device returns JTK_ERR_NO_DEVICE which is raised as JtkError.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
# JTK_LC_LIB: an experiment build of the library (scripts/*probe*.sh); the product never sets it
LIB_PATH = os.environ.get("JTK_LC_LIB") or os.path.join(PKG_DIR, "_build", "libjtk_lc.so")
SYNTH_LIB_PATH = os.path.join(PKG_DIR, "_build", "libjtk_synth.so")

NUM_ROW = 14
OP_MATCH, OP_MISMATCH, OP_INS, OP_DEL = 0, 1, 2, 3   # enum jtk_op (jtk_lc.h) == kiley::Op
GAINS_MAX_HOMOP = 8
K_COUNT = 4
KERNEL_NAMES = ("phmm", "polish", "filter", "mcmc")


class JtkError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"jtk_lc status {status}: {msg}")
        self.status = status


class Hmm(C.Structure):
    """jtk_hmm_t == definitions::HMMParam (definitions/src/lib.rs:101-126)."""
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


class SynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("tmpl_len", C.c_uint32), ("n_haps", C.c_uint32),
                ("reads_per_hap", C.c_uint32), ("min_variants", C.c_uint32), ("divergence", C.c_double),
                ("err_sub", C.c_double), ("err_ins", C.c_double), ("err_del", C.c_double),
                ("tmpl_err", C.c_double)]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
                ("kernel_ms", C.c_double * K_COUNT), ("kernel_launches", C.c_uint32 * K_COUNT),
                ("chain_lds_bytes", C.c_uint32 * 2)]


CHUNK_DT = np.dtype([("chunk_id", "<u8"), ("copy_num", "<u4"), ("n_reads", "<u4"), ("tmpl_off", "<u8"),
                     ("tmpl_len", "<u8"), ("read_first", "<u8")])
RESULT_DT = np.dtype([("score", "<f8"), ("cluster_num", "<u4"), ("status", "<i4"), ("polish_rounds", "<u4"),
                      ("n_variants", "<u4")])
FEATURE_CHUNK_DT = np.dtype([("chunk_id", "<u8"), ("copy_num", "<u4"), ("n_reads", "<u4"), ("dim", "<u4"),
                             ("reserved", "<u4"), ("var_off", "<u8"), ("vt_off", "<u8"), ("read_first", "<u8"),
                             ("local_coverage", "<f8")])

CC_NODE_DT = np.dtype([("chunk", "<u8"), ("cluster", "<u8"), ("is_forward", "<u4"), ("post_len", "<u4"), ("post_off", "<u8")])
CC_CHUNK_DT = np.dtype([("id", "<u8"), ("cluster_num", "<u4"), ("copy_num", "<u4"), ("score", "<f8")])

# every symbol declared in include/jtk_lc.h (tests check the library exports them)
EXPORTED_SYMBOLS = (
    "jtk_lc_cluster_chunks", "jtk_lc_cluster_chunks_multi", "jtk_lc_cluster_polished", "jtk_lc_polish_chunks", "jtk_lc_modification_table",
    "jtk_lc_cluster_features", "jtk_lc_estimate_gains", "jtk_lc_estimate_minimum_gain", "jtk_lc_fit_model", "jtk_lc_correct_clustering", "jtk_lc_trim_cache", "jtk_lc_pileup_sort_key", "jtk_lc_normalize_pileup", "jtk_lc_strerror",
    "jtk_lc_last_error", "jtk_lc_version", "jtk_lc_device_ok", "jtk_lc_last_timing",
    "jtk_lc_session_create", "jtk_lc_session_run", "jtk_lc_session_fetch", "jtk_lc_session_destroy", "jtk_lc_session_trace",
)
SYNTH_SYMBOLS = ("jtk_synth_pileup",)


def u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def f64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


_lib = None


def lib():
    """Load libjtk_lc.so; raise (never fall back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). jtk_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, sz, u64, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int
    PU8, PU64, PU32, PD = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                           C.POINTER(C.c_double))
    PP = C.POINTER(Params)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("jtk_lc_cluster_chunks", i32, PP, sz, vp, PU8, PU8, PU64, PU8, PU64, PU8, PU32, PD, u32, vp, PU8,
        PU64, u64, PU8, PU64, u64, i32)
    sig("jtk_lc_cluster_chunks_multi", i32, PP, sz, vp, PU8, PU8, PU64, PU8, PU64, PU8, PU32, PD, u32, vp, PU8,
        PU64, u64, PU8, PU64, u64, C.POINTER(C.c_int), sz)
    sig("jtk_lc_cluster_polished", i32, PP, sz, vp, PU8, PU8, PU64, PU8, PU64, PU8, PU32, PD, u32, vp, i32)
    sig("jtk_lc_polish_chunks", i32, PP, sz, vp, PU8, PU8, PU64, PU8, PU64, PU8, u32, u32, u32, PU8, PU64, u64, PU8, PU64,
        u64, vp, i32)
    sig("jtk_lc_modification_table", i32, PP, PU8, u64, u32, PU8, PU64, PU8, PU64, PU8, PD, PD, i32)
    sig("jtk_lc_estimate_gains", i32, C.POINTER(Hmm), C.POINTER(Hmm), u64, u32, u32, u32, C.POINTER(Gains), i32)
    sig("jtk_lc_estimate_minimum_gain", i32, C.POINTER(Hmm), C.POINTER(Hmm), u64, u32, u32, u32, u32, PD, i32)
    sig("jtk_lc_fit_model", i32, PP, sz, vp, PU8, PU8, PU64, PU8, PU64, PU8, u32, C.POINTER(Hmm), C.POINTER(Hmm), i32)
    sig("jtk_lc_correct_clustering", i32, sz, PU64, PU64, vp, PD, sz, vp, sz, PU64, C.c_double, C.c_double, PU64, PU8, i32)
    sig("jtk_lc_trim_cache", i32, i32)
    sig("jtk_lc_cluster_features", i32, PP, sz, vp, PD, PU32, PU32, PD, u32, vp, i32)
    sig("jtk_lc_pileup_sort_key", i32, PU8, u64, PU8, u64, PU8, u64, PU64)
    sig("jtk_lc_normalize_pileup", i32, u32, u32, PU32, PD, u32)
    sig("jtk_lc_strerror", C.c_char_p, i32)
    sig("jtk_lc_last_error", C.c_char_p)
    sig("jtk_lc_version", i32)
    sig("jtk_lc_device_ok", i32, i32)
    sig("jtk_lc_last_timing", i32, C.POINTER(Timing))
    sig("jtk_lc_session_create", i32, PP, sz, vp, PU8, PU8, PU64, PU8, PU64, PU8, u32, i32,
        C.POINTER(vp))
    sig("jtk_lc_session_run", i32, vp, i32)
    sig("jtk_lc_session_fetch", i32, vp, PU32, PD, vp, PU8, PU64, u64, PU8, PU64, u64)
    sig("jtk_lc_session_destroy", i32, vp)
    sig("jtk_lc_session_trace", i32, vp, sz, C.c_char_p, sz, C.POINTER(sz))
    _lib = L
    return L


_synth = None


def synth_lib():
    """libjtk_synth.so: synthetic inputs for bench.py and the tests (not part of the product library)."""
    global _synth
    if _synth is None:
        if not os.path.exists(SYNTH_LIB_PATH):
            raise ImportError(f"{SYNTH_LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(SYNTH_LIB_PATH)
        PU8, PU64, PU32 = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        L.jtk_synth_pileup.restype = C.c_int
        L.jtk_synth_pileup.argtypes = [C.POINTER(SynthCfg), PU8, C.c_uint64, PU64, PU8, C.c_uint64, PU64, PU8,
                                       C.c_uint64, PU64, PU8, PU32]
        _synth = L
    return _synth


def check(status):
    if status != 0:
        L = lib()
        msg = L.jtk_lc_strerror(status).decode()
        extra = L.jtk_lc_last_error().decode()
        raise JtkError(status, msg + (": " + extra if extra else ""))


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
