"""The oracle's Baum-Welch refit (oracle/model_fit.c; own specification of kiley's fit, driver after model_tune.rs:96-156):
identities every set of expected counts satisfies, and the direction the fit moves a model."""
import ctypes as C

import numpy as np

import helpers
import oracle_ffi as O


def counts_of(hmm, tmpl, read, ops, radius):
    cnt = np.zeros(45)
    lk = O.lib().jo_phmm_counts(C.byref(hmm), O.u8p(tmpl), len(tmpl), O.u8p(read), len(read), O.u8p(ops), len(ops), radius,
                                O.f64p(cnt))
    return lk, cnt


def test_expected_counts_satisfy_the_path_identities(oracle):
    b, cfg, p = helpers.small_batch(config="ont_noisy", n_chunks=1, tmpl_len=300, reads_per_hap=3)
    t = b.template(0)
    for r in b.chunk_reads(0):
        rd, op = b.read(r), b.read_ops(r)
        lk, cnt = counts_of(O.default_hmm(), t, rd, op, 12)
        assert lk == O.likelihood(O.default_hmm(), t, rd, op, 12)           # the same forward sweep
        tr = cnt[:9].reshape(3, 3)                                            # [from M, I, D][to M, I, D]
        em, ei = cnt[9:25].sum(), cnt[25:45].sum()
        assert abs(tr[:, 0].sum() - em) < 1e-9 and abs(tr[:, 1].sum() - ei) < 1e-9   # a visit is entered exactly once
        assert abs(em + tr[:, 2].sum() - len(t)) < 1e-9                       # every template base: Match or Del
        assert abs(em + ei - len(rd)) < 1e-9                                  # every read base: Match or Ins
        assert (cnt >= 0).all()
    # ops that do not consume the sequences: no counts
    lk, cnt = counts_of(O.default_hmm(), t, b.read(0), b.read_ops(0)[:-2], 12)
    assert lk <= -1e299 and not cnt.any()


def test_mstep_normalises_rows_and_keeps_rows_without_mass(oracle):
    old = O.default_hmm()
    cnt = np.zeros(45)
    cnt[0:3] = [90, 6, 4]            # Mat row
    cnt[6:9] = [3, 0, 1]             # Del row; the Ins row has no mass
    cnt[9:13] = [50, 2, 1, 1]        # emissions of template base A only
    new = O.Hmm()
    O.lib().jo_fit_mstep(C.byref(old), O.f64p(cnt), C.byref(new))
    assert (new.mat_mat, new.mat_ins, new.mat_del) == (0.9, 0.06, 0.04)
    assert (new.del_mat, new.del_ins, new.del_del) == (0.75, 0.0, 0.25)
    assert (new.ins_mat, new.ins_ins, new.ins_del) == (old.ins_mat, old.ins_ins, old.ins_del)
    assert list(new.mat_emit)[:4] == [50 / 54, 2 / 54, 1 / 54, 1 / 54] and list(new.mat_emit)[4:] == list(old.mat_emit)[4:]
    assert list(new.ins_emit) == list(old.ins_emit)


def test_fit_moves_the_model_towards_the_read_error_rates(oracle):
    """reads simulated with 5 % / 5 % / 5 % errors: ten rounds (TRAIN_ROUND, model_tune.rs:95) leave a model whose Mat row is
    far from the 0.97 / 0.01 / 0.01 it started from, with proper probability rows, and a higher likelihood of the data"""
    b, cfg, p = helpers.small_batch(config="ont_noisy", n_chunks=2, tmpl_len=250, reads_per_hap=5, divergence=0.0)
    po = helpers.oracle_params(p)
    rc, f, r = O.fit_model(po, b, rounds=3)
    assert rc == 0
    for h in (f, r):
        for row in ((h.mat_mat, h.mat_ins, h.mat_del), (h.ins_mat, h.ins_ins, h.ins_del), (h.del_mat, h.del_ins, h.del_del)):
            assert abs(sum(row) - 1.0) < 1e-12 and min(row) >= 0
        for x in range(4):
            assert abs(sum(list(h.mat_emit)[4 * x:4 * x + 4]) - 1.0) < 1e-12
        assert 0.03 < h.mat_ins < 0.12 and 0.03 < h.mat_del < 0.12 and h.mat_emit[0] < 0.97
    before = sum(O.likelihood(po.forward if b.strand[g] else po.reverse, b.template(0), b.read(g), b.read_ops(g), 8)
                 for g in b.chunk_reads(0))
    after = sum(O.likelihood(f if b.strand[g] else r, b.template(0), b.read(g), b.read_ops(g), 8) for g in b.chunk_reads(0))
    assert after > before
