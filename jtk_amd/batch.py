"""Flat batch layout of include/jtk_lc.h (what a Rust host flattens `pileups` into, mod.rs:63-72)."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import ffi

# Gains for HMMParam::default() on both strands, i.e. what
# `estimate_gain_default(&hmm)` (likelihood_gains.rs:186-192) hands to the stage. These are INPUTS at the
# boundary (a jtk host computes them with kiley); the numbers below were produced once by the CPU
# restatement oracle/likelihood_gains.c (tests/golden/make_gains.py) and are data, not code.
DEFAULT_GAINS = {
    "subst": [(4.564428406624671, 0.02), (4.561695470515055, 0.02), (4.556543808195073, 0.02)],
    "deletions": [(4.56443060675976, 0.04), (3.8914601411150223, 0.04), (3.4928478696557903, 0.04)],
    "insertions": [(5.221977240299079, 0.04), (5.909607854065992, 0.02), (5.899462289114112, 0.02)],
}


def default_params(haploid_coverage, band_frac=0.03, gains=None, hmm=None):
    p = ffi.Params()
    p.forward = hmm if hmm is not None else ffi.default_hmm()
    p.reverse = hmm if hmm is not None else ffi.default_hmm()
    g = gains if gains is not None else DEFAULT_GAINS
    p.gains.max_homopolymer_len = len(g["subst"])
    for name in ("subst", "deletions", "insertions"):
        arr = getattr(p.gains, name)
        for i, (gain, prob) in enumerate(g[name]):
            arr[i].gain = gain
            arr[i].prob = prob
    p.haploid_coverage = float(haploid_coverage)
    p.band_frac = float(band_frac)
    return p


@dataclass
class Batch:
    """Inputs of jtk_lc_cluster_chunks. Reads of chunk c are read_first .. read_first+n_reads."""
    chunks: np.ndarray            # ffi.CHUNK_DT [n_chunks]
    tmpl_bases: np.ndarray        # u8
    read_bases: np.ndarray        # u8
    read_off: np.ndarray          # u64 [n_reads+1]
    ops: np.ndarray               # u8 (0=Match 1=Mismatch 2=Ins 3=Del)
    ops_off: np.ndarray           # u64 [n_reads+1]
    strand: np.ndarray            # u8 [n_reads] (1 = forward)
    truth: np.ndarray = field(default=None)  # u32 [n_reads] generating haplotype (synthetic only)

    @property
    def n_chunks(self):
        return len(self.chunks)

    @property
    def n_reads(self):
        return len(self.strand)

    @property
    def post_stride(self):
        return int(max(1, self.chunks["copy_num"].max())) if len(self.chunks) else 1

    def chunk_reads(self, c):
        ch = self.chunks[c]
        return range(int(ch["read_first"]), int(ch["read_first"]) + int(ch["n_reads"]))

    def template(self, c):
        ch = self.chunks[c]
        return self.tmpl_bases[int(ch["tmpl_off"]):int(ch["tmpl_off"]) + int(ch["tmpl_len"])]

    def read(self, r):
        return self.read_bases[int(self.read_off[r]):int(self.read_off[r + 1])]

    def read_ops(self, r):
        return self.ops[int(self.ops_off[r]):int(self.ops_off[r + 1])]

    def subset(self, idx):
        """Batch made of the chunks `idx` (re-packed)."""
        return pack([(int(self.chunks[c]["chunk_id"]), int(self.chunks[c]["copy_num"]), self.template(c),
                      [self.read(r) for r in self.chunk_reads(c)],
                      [self.read_ops(r) for r in self.chunk_reads(c)],
                      [int(self.strand[r]) for r in self.chunk_reads(c)],
                      None if self.truth is None else [int(self.truth[r]) for r in self.chunk_reads(c)])
                     for c in idx])

    def algorithmic_bytes(self, k_per_chunk=None):
        """packed 4-bit bases, 2-bit ops, u32 labels, f64 log-posteriors."""
        total = 0.0
        for c in range(self.n_chunks):
            ch = self.chunks[c]
            n, L = int(ch["n_reads"]), int(ch["tmpl_len"])
            k = int(ch["copy_num"]) if k_per_chunk is None else int(k_per_chunk[c])
            r0, r1 = int(ch["read_first"]), int(ch["read_first"]) + n
            read_b = int(self.read_off[r1] - self.read_off[r0]) / 2
            ops_b = int(self.ops_off[r1] - self.ops_off[r0]) / 4
            total += read_b + ops_b + L / 2 + n * (4 + 8 * k) + L / 2 + ops_b + 16
        return total


def pileup_sort(tmpl, reads, ops):
    """Order of pileup_nodes (mod.rs:45-51): stable sort by #columns != '|' (Node::recover)."""
    L = ffi.lib()
    keys = []
    for rd, op in zip(reads, ops):
        key = C.c_uint64(0)
        ffi.check(L.jtk_lc_pileup_sort_key(ffi.u8p(tmpl), len(tmpl), ffi.u8p(rd), len(rd), ffi.u8p(op), len(op),
                                           C.byref(key)))
        keys.append(key.value)
    return sorted(range(len(reads)), key=lambda i: keys[i])  # Python's sort is stable


def pack(pileups):
    """pileups: iterable of (chunk_id, copy_num, tmpl u8[], reads [u8[]], ops [u8[]], strands [int], truth|None)."""
    chunks, tb, rb, ob, ro, oo, st, tr = [], [], [], [], [0], [0], [], []
    toff = 0
    nread = 0
    have_truth = True
    for cid, cn, tmpl, reads, ops, strands, truth in pileups:
        chunks.append((cid, cn, len(reads), toff, len(tmpl), nread))
        tb.append(np.asarray(tmpl, dtype=np.uint8))
        toff += len(tmpl)
        for i, (r, o) in enumerate(zip(reads, ops)):
            rb.append(np.asarray(r, dtype=np.uint8))
            ob.append(np.asarray(o, dtype=np.uint8))
            ro.append(ro[-1] + len(r))
            oo.append(oo[-1] + len(o))
            st.append(1 if strands[i] else 0)
        if truth is None:
            have_truth = False
        else:
            tr.extend(truth)
        nread += len(reads)
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.uint8)  # noqa: E731
    return Batch(chunks=np.array(chunks, dtype=ffi.CHUNK_DT), tmpl_bases=cat(tb), read_bases=cat(rb),
                 read_off=np.array(ro, dtype=np.uint64), ops=cat(ob), ops_off=np.array(oo, dtype=np.uint64),
                 strand=np.array(st, dtype=np.uint8),
                 truth=np.array(tr, dtype=np.uint32) if have_truth else None)
