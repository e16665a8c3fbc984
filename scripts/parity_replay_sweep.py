"""Diagnostic (GPU box): randomized parity of the modification table (phmm_kernel's checkpoint + replay path, phmm_wide_kernel
above radius 30) against the CPU oracle: random template lengths 1,000 .. 2,150 (band radius 15 .. 32), reads with random
substitutions, single indels and indel runs, random strands; lk and every table entry bit for bit.
`python scripts/parity_replay_sweep.py [n_cases] [seed]`"""
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
from jtk_amd import api, batch as jb  # noqa: E402

spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
p = jb.default_params(haploid_coverage=25.0)
po = helpers.oracle_params(p)
bad = 0
for case in range(n_cases):
    L = int(rng.integers(1000, 2150))
    tmpl = rng.integers(0, 4, L).astype(np.uint8)
    n = int(rng.integers(2, 6))
    reads, ops = [], []
    for r in range(n):
        runs = {}
        style = int(rng.integers(0, 3))
        if style >= 1:   # single indels everywhere (3 % each)
            for pos in np.flatnonzero(rng.random(L) < 0.06):
                runs[int(pos)] = 1 if rng.random() < 0.5 else -1
        if style == 2:   # and runs
            pos = int(rng.integers(3, 60))
            while pos < L - 20:
                k = int(rng.integers(2, 14))
                runs[pos] = k if rng.random() < 0.5 else -k
                pos += k + int(rng.integers(1, 200))
        rd, op = m.indel_run_read(rng, tmpl, runs)
        reads.append(rd)
        ops.append(op)
    strands = [int(x) for x in rng.integers(0, 2, n)]
    t = m.ACGT[tmpl]
    tab, lk = api.modification_table(p, t, reads, ops, strands)
    rb, ob = np.concatenate(reads), np.concatenate(ops)
    ro = np.zeros(n + 1, np.uint64)
    oo = np.zeros(n + 1, np.uint64)
    ro[1:] = np.cumsum([len(x) for x in reads])
    oo[1:] = np.cumsum([len(x) for x in ops])
    otab = np.zeros_like(tab)
    olk = np.zeros(n)
    O.lib().jo_modification_table(C.byref(po), O.u8p(t), L, n, O.u8p(rb), O.u64p(ro), O.u8p(ob), O.u64p(oo),
                                  O.u8p(np.array(strands, np.uint8)), O.f64p(otab), O.f64p(olk))
    ok = np.array_equal(helpers.bits(lk), helpers.bits(olk)) and np.array_equal(helpers.bits(tab), helpers.bits(otab))
    print(case, "L", L, "radius", int(np.ceil(L * p.band_frac)) // 2 if hasattr(p, "band_frac") else "?", "reads", n,
          "T mod 64", [(L + len(x)) % 64 for x in reads], "OK" if ok else "MISMATCH")
    bad += 0 if ok else 1
print("mismatching cases", bad, "of", n_cases)
