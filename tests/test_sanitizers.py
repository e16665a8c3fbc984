"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the CPU-side code (GPU sanitizers are not available on this pool):
the oracle (`make -C oracle asan`) through a full small pile-up, the recursive split, the chain and the exact comparator,
and the host-only entry points of the product library (jtk_amd/csrc/host_api.cpp) built with the same flags.  Each runs
in a child process with libasan preloaded; any report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def libasan():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def run_child(code, env_extra):
    env = dict(os.environ, **env_extra)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:exitcode=77"   # CPython itself "leaks" at exit
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["LD_PRELOAD"] = libasan()
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, \
        (r.returncode, r.stderr[-3000:])
    return r.stdout


@pytest.mark.skipif(libasan() is None, reason="libasan not found")
def test_oracle_under_asan_ubsan(jtk_lib):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    code = r'''
import sys, os
sys.path.insert(0, "tests")
import numpy as np
import oracle_ffi as O
O.ORACLE_SO = os.path.join(O.ORACLE_DIR, "_build", "libjtk_oracle_asan.so")
O.build = lambda force=False: O.ORACLE_SO
import helpers
from jtk_amd import batch as jb
# full path, small pile-ups (polish + tables + filter + chain), then the copy_num >= 8 split
b, cfg, p = helpers.small_batch(n_chunks=2, tmpl_len=250, reads_per_hap=5)
o = O.cluster_chunks(helpers.oracle_params(p), b, n_threads=2)
assert o["rc"] == 0
b, cfg, p = helpers.small_batch(config="ont_4copy", n_chunks=1, tmpl_len=300, reads_per_hap=5, n_haps=8, copy_num=9,
                                divergence=2e-2, min_variants=3)
o = O.cluster_chunks(helpers.oracle_params(p), b, n_threads=1)
assert o["rc"] == 0
# the exact comparator and a chunk that fails where the reference asserts
x = np.ascontiguousarray(np.random.default_rng(1).normal(size=(9, 4)))
asn = np.zeros(9, dtype=np.uintp); gain = np.zeros((9, 3))
O.lib().jo_cluster_filtered_variants_exact(O.f64p(x), 9, 4, 3, O.szp(asn), O.f64p(gain))
# the cross-chunk correction (oracle/correction.c)
prob = helpers.correction_problem(3, n_chunks=5, n_reads=30, wrong=0.05)
rc, *_ = O.correct_clustering(prob["read_id"], prob["node_off"], prob["nodes"], prob["posteriors"], prob["chunks"].copy(),
                              np.arange(5, dtype=np.uint64), 20.0, 1.0, 0)
assert rc == 0
# the trace! rows of a chunk (oracle/pseudo_mcmc.c: trace_row), into a buffer that holds them and into one that does not
b, cfg, p = helpers.small_batch(n_chunks=1, tmpl_len=250, reads_per_hap=5)
out, rows = O.trace_chunk(helpers.oracle_params(p), b, 0)
assert out["rc"] == 0 and rows and rows[0].startswith("TOTAL")
import ctypes as C
buf = C.create_string_buffer(16)
t = O.Trace(C.cast(buf, C.c_void_p), len(buf), 0)
O.lib().jo_trace_set(C.byref(t))
O._cluster_chunks_live(helpers.oracle_params(p), b, n_threads=1)
O.lib().jo_trace_set(None)
assert t.len == sum(len(r) + 1 for r in rows) > 16 and buf.raw[:6] == b"TOTAL\t"
print("ok")
'''
    assert "ok" in run_child(code, {})


@pytest.mark.skipif(libasan() is None, reason="libasan not found")
def test_host_entry_points_under_asan_ubsan(tmp_path):
    so = tmp_path / "libjtk_host_asan.so"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "jtk_amd", "csrc", "host_api.cpp"),
                           os.path.join(ROOT, "jtk_amd", "csrc", "synth.cpp"), "-o", str(so)])
    code = r'''
import ctypes as C, os, sys
import numpy as np
L = C.CDLL(os.environ["JTK_HOST_ASAN_SO"])
PU8, PU32, PD = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_double)
rng = np.random.default_rng(3)
for n, k in [(1, 1), (25, 3), (40, 7), (0, 2)]:
    lab = rng.integers(0, k, n).astype(np.uint32)
    post = rng.normal(size=(n, k + 2))
    rc = L.jtk_lc_normalize_pileup(n, k, lab.ctypes.data_as(PU32), post.ctypes.data_as(PD), k + 2)
    assert rc == 0, rc
lab = np.array([0, 5], np.uint32); post = np.zeros((2, 2))
assert L.jtk_lc_normalize_pileup(2, 2, lab.ctypes.data_as(PU32), post.ctypes.data_as(PD), 2) == -1   # label out of range
t = np.frombuffer(b"ACGTACGT", np.uint8).copy(); r = np.frombuffer(b"ACTTAGT", np.uint8).copy()
ops = np.array([0, 0, 1, 0, 0, 3, 0, 0], np.uint8)
key = C.c_uint64()
assert L.jtk_lc_pileup_sort_key(t.ctypes.data_as(PU8), C.c_uint64(8), r.ctypes.data_as(PU8), C.c_uint64(7), ops.ctypes.data_as(PU8), C.c_uint64(8), C.byref(key)) == 0
assert L.jtk_lc_pileup_sort_key(t.ctypes.data_as(PU8), C.c_uint64(8), r.ctypes.data_as(PU8), C.c_uint64(7), ops.ctypes.data_as(PU8), C.c_uint64(5), C.byref(key)) == -5
L.jtk_lc_strerror.restype = C.c_char_p
for s in range(-9, 2):
    assert L.jtk_lc_strerror(s)
print("ok")
'''
    assert "ok" in run_child(code, {"JTK_HOST_ASAN_SO": str(so)})
