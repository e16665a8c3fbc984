"""Diagnostic (GPU box): one feature problem of scripts/parity_sweep.py (seed, chunk) against the oracle."""
import ctypes as C
import sys
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import api, ffi, batch as jb  # noqa: E402

seed, chunk = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
specs = []
for _ in range(14):
    cn = int(rng.choice([2, 2, 2, 2, 3, 4]))
    n = int(rng.integers(4, 128)) if cn == 2 else int(rng.integers(6, 90))
    d = int(rng.integers(1, 9)) if cn == 2 else int(rng.integers(1, 3 * cn + 1))
    specs.append((n, d, int(rng.integers(1, cn + 1)), cn))
p = jb.default_params(haploid_coverage=float(rng.choice([8.0, 15.0, 30.0])))
prng = np.random.default_rng(seed)  # run_features_both draws the problems in order from default_rng(seed)
for i, (n, dim, k_true, copy_num) in enumerate(specs[:chunk + 1]):
    x, vt, lab = helpers.random_feature_problem(prng, n, dim, k_true)
ch = np.zeros(1, dtype=ffi.FEATURE_CHUNK_DT)
ch[0] = (1000 + 17 * chunk, copy_num, n, dim, 0, 0, 0, 0, n / copy_num)
var = np.ascontiguousarray(x.ravel())
vts = np.ascontiguousarray(vt.ravel().astype(np.uint32))
stride = max(2, copy_num)
dev = api.cluster_features(p, ch, var, vts, stride)
po = helpers.oracle_params(p)
olab = np.zeros(n, np.uint32)
opost = np.zeros((n, stride))
ores = np.zeros(1, dtype=ffi.RESULT_DT)
assert O.lib().jo_cluster_features(C.byref(po), 1, ch.ctypes.data, O.f64p(var), O.u32p(vts), O.u32p(olab), O.f64p(opost), stride,
                                   ores.ctypes.data, 0) == 0
ok = np.array_equal(dev["label"], olab) and helpers.bits(dev["result"]["score"])[0] == helpers.bits(ores["score"])[0]
print("REPRO", seed, chunk, specs[chunk], "OK" if ok else "MISMATCH", dev["result"]["score"][0], ores["score"][0], flush=True)
