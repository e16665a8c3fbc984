"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the public headers
declare, host helpers agree with the oracle, the product never reaches for the oracle, and compute calls
fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers
import oracle_ffi as O
from jtk_amd import api, batch as jb, ffi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(jtk_lib):
    def declared_in(hdr):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        return set(re.findall(r"\b(jtk_(?:lc|synth)_[a-z_0-9]+)\s*\(", text))
    declared = declared_in("jtk_lc.h")
    assert declared == set(ffi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(jtk_lib, name), name
    # the synthetic-input generator is test/bench infrastructure: its own library, not the product's
    assert declared_in("jtk_synth.h") == set(ffi.SYNTH_SYMBOLS)
    for name in ffi.SYNTH_SYMBOLS:
        assert hasattr(ffi.synth_lib(), name) and not hasattr(jtk_lib, name), name
    assert jtk_lib.jtk_lc_version() == 2
    assert jtk_lib.jtk_lc_strerror(-5).decode().startswith("alignment ops")


def test_library_exports_nothing_but_the_declared_entry_points():
    """the library is built with -fvisibility=hidden: what `nm -D` lists as defined are the entry points of include/jtk_lc.h and
    the diagnostic ones of include/jtk_lc_debug.h -- no helper of the implementation (a Rust binary links these by name)"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", ffi.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    names = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-2] in "TDBRW"}
    debug = set(re.findall(r"\b(jtk_lc_debug_[a-z_0-9]+)\s*\(", re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "jtk_lc_debug.h")).read(), flags=re.S)))
    assert debug, "jtk_lc_debug.h declares the diagnostic entry points"
    assert names == set(ffi.EXPORTED_SYMBOLS) | debug, sorted(names ^ (set(ffi.EXPORTED_SYMBOLS) | debug))


def test_struct_layouts_match_between_bindings():
    assert C.sizeof(ffi.Params) == C.sizeof(O.Params) == 2 * 45 * 8 + 8 + 3 * 8 * 16 + 16
    assert ffi.CHUNK_DT.itemsize == 40 and ffi.RESULT_DT.itemsize == 24 and ffi.FEATURE_CHUNK_DT.itemsize == 56


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "jtk_amd")
    for dirpath, _, files in os.walk(pkg):
        if "_build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_ffi" not in text and "jtk_oracle" not in text and "libjtk_oracle" not in text, f


def test_pileup_sort_key_matches_oracle(jtk_lib, oracle):
    b, cfg, _ = helpers.small_batch(n_chunks=2, tmpl_len=300, reads_per_hap=6)
    Lo = oracle.lib()
    for c in range(b.n_chunks):
        t = b.template(c)
        keys = []
        for r in b.chunk_reads(c):
            rd, op = b.read(r), b.read_ops(r)
            k = C.c_uint64()
            assert jtk_lib.jtk_lc_pileup_sort_key(ffi.u8p(t), len(t), ffi.u8p(rd), len(rd), ffi.u8p(op), len(op),
                                                  C.byref(k)) == 0
            assert k.value == Lo.jo_pileup_sort_key(O.u8p(t), len(t), O.u8p(rd), len(rd), O.u8p(op), len(op))
            keys.append(k.value)
        assert keys == sorted(keys)          # make_batch delivers reads in pileup_nodes order
    bad = np.array([0, 0, 3], dtype=np.uint8)
    k = C.c_uint64()
    t = O.seq(b"ACG")
    assert jtk_lib.jtk_lc_pileup_sort_key(ffi.u8p(t), 3, ffi.u8p(t), 3, ffi.u8p(bad), 3, C.byref(k)) == -5


def test_normalize_pileup_matches_oracle(jtk_lib, oracle):
    rng = np.random.default_rng(5)
    Lo = oracle.lib()
    for k in (1, 2, 3, 4, 7):
        for trial in range(20):
            n = 25
            lab = rng.integers(0, k, n).astype(np.uint32)
            if trial % 3 == 0 and k > 2:
                lab[lab == 1] = 0     # an empty cluster and ties
            post = rng.normal(size=(n, k + 1))
            l1, p1 = api.normalize_pileup(lab.copy(), post.copy(), k)
            l2 = lab.astype(np.uint64)
            p2 = post.copy()
            Lo.jo_normalize_pileup(n, k, O.u64p(l2), O.f64p(p2), k + 1)
            assert l1.tolist() == l2.tolist()
            assert np.array_equal(p1, p2)
            counts = np.bincount(l1, minlength=k)
            assert all(counts[i] >= counts[i + 1] for i in range(k - 1))


def test_synth_is_deterministic_and_well_formed(jtk_lib):
    b1, cfg, _ = helpers.small_batch(n_chunks=2, tmpl_len=500, reads_per_hap=8)
    b2, _, _ = helpers.small_batch(n_chunks=2, tmpl_len=500, reads_per_hap=8)
    assert np.array_equal(b1.read_bases, b2.read_bases) and np.array_equal(b1.ops, b2.ops)
    for c in range(b1.n_chunks):
        t = b1.template(c)
        assert set(np.unique(t)) <= set(b"ACGT")
        for r in b1.chunk_reads(c):
            o = b1.read_ops(r)
            assert (o != 2).sum() == len(t) and (o != 3).sum() == len(b1.read(r))


def test_algorithmic_bytes_follow_the_survey_formula(jtk_lib):
    """B(N, L, k) = N L/2 + N 1.05L/4 + L/2 + N (4 + 8k) + L/2 + N 1.05L/4 + 16; B(60, 2000, 2) =
    126,216 B.  Batch.algorithmic_bytes counts the ACTUAL read and ops lengths (the 1.05 L of the formula is their
    nominal value), so the check is exact on the actual terms and within 3 % of the nominal figure."""
    b, cfg = synth.make_batch("ont_diploid", 3)
    got = b.algorithmic_bytes()
    exact = 0.0
    for c in range(b.n_chunks):
        rr = list(b.chunk_reads(c))
        n, L = len(rr), len(b.template(c))
        assert n == 60
        reads = sum(len(b.read(r)) for r in rr)
        ops = sum(len(b.read_ops(r)) for r in rr)
        exact += reads / 2 + ops / 4 + L / 2 + n * (4 + 8 * 2) + L / 2 + ops / 4 + 16
    assert got == exact
    assert abs(got / b.n_chunks - 126216) < 0.03 * 126216
    one = b.subset([0])
    assert one.algorithmic_bytes(k_per_chunk=[1]) == one.algorithmic_bytes() - 60 * 8


def test_compute_fails_loudly_without_gpu(jtk_lib):
    """No silent CPU fallback: on a box without a gfx950 device every compute entry returns NO_DEVICE."""
    if jtk_lib.jtk_lc_device_ok(0):
        pytest.skip("a GPU is present")
    b, cfg, p = helpers.small_batch(n_chunks=1, tmpl_len=200, reads_per_hap=4)
    with pytest.raises(ffi.JtkError) as e:
        api.cluster_chunks(p, b)
    assert e.value.status == -2
    with pytest.raises(ffi.JtkError):
        api.Session(p, b)
    with pytest.raises(ffi.JtkError) as e:
        api.estimate_gains(p.forward, p.reverse)
    assert e.value.status == -2


def test_invalid_arguments_are_rejected(jtk_lib):
    b, cfg, p = helpers.small_batch(n_chunks=1, tmpl_len=200, reads_per_hap=4)
    h = C.c_void_p()
    assert jtk_lib.jtk_lc_session_create(None, 1, b.chunks.ctypes.data, ffi.u8p(b.tmpl_bases), ffi.u8p(b.read_bases),
                                         ffi.u64p(b.read_off), ffi.u8p(b.ops), ffi.u64p(b.ops_off),
                                         ffi.u8p(b.strand), 2, 0, C.byref(h)) == -1


def test_polish_chunks_rejects_null_arguments(jtk_lib):
    """jtk_lc_polish_chunks validates before it dereferences (params == NULL or chunks == NULL used to be a crash)"""
    b, cfg, p = helpers.small_batch(n_chunks=1, tmpl_len=200, reads_per_hap=4)
    out = api._outputs(b)
    args = lambda params, chunks: (params, 1, chunks, ffi.u8p(b.tmpl_bases), ffi.u8p(b.read_bases), ffi.u64p(b.read_off),  # noqa: E731
                                  ffi.u8p(b.ops), ffi.u64p(b.ops_off), ffi.u8p(b.strand), 0, 0, 0, ffi.u8p(out["cons"]),
                                  ffi.u64p(out["cons_off"]), len(out["cons"]), ffi.u8p(out["ops_out"]), ffi.u64p(out["ops_out_off"]),
                                  len(out["ops_out"]), out["result"].ctypes.data, 0)
    assert jtk_lib.jtk_lc_polish_chunks(*args(None, b.chunks.ctypes.data)) == -1
    assert jtk_lib.jtk_lc_polish_chunks(*args(C.byref(p), None)) == -1
