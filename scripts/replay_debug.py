"""Diagnostic (GPU box): the indel-run parity case of tests/test_gpu_parity.py, read by read: which reads differ from the oracle,
from which template row on, in which table rows.  usage: replay_debug.py <tmpl_len> <seed>   (JTK_LC_LIB picks the library)"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import api, batch as jb, ffi  # noqa: E402

spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
tmpl_len, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
p = jb.default_params(haploid_coverage=25.0)
tmpl = rng.integers(0, 4, tmpl_len).astype(np.uint8)
reads, ops, allruns = [], [], []
for r in range(6):
    runs, pos = {}, int(rng.integers(3, 40))
    while pos < tmpl_len - 20:
        k = int(rng.integers(3, 16))
        runs[pos] = k if rng.random() < 0.6 else -k
        pos += k + int(rng.integers(1, 90))
    if r == 0:
        runs = {}
    allruns.append(dict(runs))
    rd, op = m.indel_run_read(rng, tmpl, runs)
    reads.append(rd)
    ops.append(op)
strands = [1, 0, 1, 0, 1, 1]
tmpl = m.ACGT[tmpl]
po = helpers.oracle_params(p)
for r in range(6):
    tab, lk = api.modification_table(p, tmpl, [reads[r]], [ops[r]], [strands[r]])
    otab = np.zeros_like(tab)
    olk = np.zeros(1)
    ro = np.array([0, len(reads[r])], np.uint64)
    oo = np.array([0, len(ops[r])], np.uint64)
    O.lib().jo_modification_table(C.byref(po), O.u8p(tmpl), tmpl_len, 1, O.u8p(reads[r]), O.u64p(ro), O.u8p(ops[r]), O.u64p(oo),
                                  O.u8p(np.array([strands[r]], np.uint8)), O.f64p(otab), O.f64p(olk))
    t = tab.reshape(tmpl_len + 1, ffi.NUM_ROW)
    o = otab.reshape(tmpl_len + 1, ffi.NUM_ROW)
    bad = np.argwhere(helpers.bits(t) != helpers.bits(o))
    print("read %d len %d T %d: lk %s; %d table entries differ" % (r, len(reads[r]), tmpl_len + len(reads[r]),
                                                                   "same" if helpers.bits(lk)[0] == helpers.bits(olk)[0] else "DIFFERS", len(bad)))
    if len(bad):
        rows = np.unique(bad[:, 0])
        print("   rows %d .. %d (%d rows), entries of the first: %s; max |diff| %.3g" % (rows[0], rows[-1], len(rows), bad[bad[:, 0] == rows[0], 1].tolist(),
              np.nanmax(np.abs(t[rows] - o[rows]))))
        print("   rows:", rows[:40].tolist())
        near = {k: v for k, v in allruns[r].items() if rows[0] - 80 <= k <= rows[-1] + 80}
        print("   runs near:", near)
