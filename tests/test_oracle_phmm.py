"""Self-consistency of the oracle's OWN-SPEC pair-HMM (kiley is absent: parity unpinned, see oracle/phmm.c):
every modification-table entry must equal the likelihood of the explicitly edited template."""
import ctypes as C

import numpy as np

import oracle_ffi as O


def make_read(rng, tm, err=0.02):
    rd = []
    for b in tm:
        u = rng.random()
        if u < err:
            continue
        if u < 2 * err:
            rd.append(b"ACGT"[rng.integers(0, 4)])
        if u < 3 * err:
            rd.append(b"ACGT"[rng.integers(0, 4)])
            continue
        rd.append(b)
    return np.array(rd, dtype=np.uint8)


def test_table_equals_edited_template_likelihood(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(7)
    Lt = 240
    tm = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, Lt)].copy()
    rd = make_read(rng, tm)
    ops = O.edit_ops(tm, rd)
    h = O.default_hmm()
    tab, lk = O.modification_table(h, tm, rd, ops, 30)
    assert lk == O.likelihood(h, tm, rd, ops, 30)
    tab = tab.reshape(-1, 14)

    def lk_of(t2):
        return L.jo_phmm_likelihood_bootstrap(C.byref(h), O.u8p(t2), len(t2), O.u8p(rd), len(rd), 60)

    worst = 0.0
    for p in [0, 1, 2, 17, 100, 200, Lt - 4, Lt - 2, Lt - 1]:
        for row in range(14):
            if row < 4:
                t2 = tm.copy()
                t2[p] = b"ACGT"[row]
            elif row < 8:
                t2 = np.concatenate([tm[:p], [b"ACGT"[row - 4]], tm[p:]]).astype(np.uint8)
            elif row < 11:
                c = row - 7
                if p + c > Lt:
                    assert tab[p, row] <= -1e299
                    continue
                t2 = np.concatenate([tm[:p], tm[p:p + c], tm[p:]]).astype(np.uint8)
            else:
                d = row - 10
                if p + d >= Lt:
                    assert tab[p, row] <= -1e299
                    continue
                t2 = np.concatenate([tm[:p], tm[p + d:]]).astype(np.uint8)
            worst = max(worst, abs(lk_of(t2) - tab[p, row]))
    assert worst < 1e-9
    # substituting a base by itself changes nothing
    codes = {65: 0, 67: 1, 71: 2, 84: 3}
    same = np.array([tab[p, codes[int(tm[p])]] - lk for p in range(Lt)])
    assert np.abs(same).max() < 1e-9
    # insertion after the last base
    t2 = np.concatenate([tm, [ord("G")]]).astype(np.uint8)
    assert abs(lk_of(t2) - tab[Lt, 4 + 2]) < 1e-9


def test_scaling_survives_long_noisy_reads(oracle):
    """15% error over 2 kbp would underflow an unscaled forward pass (~1e-900)."""
    rng = np.random.default_rng(3)
    tm = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 2000)].copy()
    rd = make_read(rng, tm, err=0.05)
    ops = O.edit_ops(tm, rd)
    lk = O.likelihood(O.default_hmm(), tm, rd, ops, 30)
    assert -6000 < lk < -800


def test_polish_recovers_truth(oracle):
    """reads drawn from a haplotype, template = haplotype + errors -> polishing returns the haplotype"""
    L = oracle.lib()
    rng = np.random.default_rng(11)
    hap = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 300)].copy()
    tm = hap.copy()
    tm[50] = ord("A") if hap[50] != ord("A") else ord("C")
    tm = np.delete(tm, 120)
    tm = np.insert(tm, 200, ord("T"))
    reads = [make_read(rng, hap, err=0.01) for _ in range(16)]
    from jtk_amd import batch as jb, ffi
    import helpers
    opss = [O.edit_ops(tm, r) for r in reads]
    b = jb.pack([(5, 1, tm, reads, opss, [1] * 16, None)])
    p = jb.default_params(8.0)
    r = O.cluster_chunks(helpers.oracle_params(p), b)
    assert r["rc"] == 0
    cons = r["cons"][int(r["cons_off"][0]):int(r["cons_off"][1])]
    assert bytes(cons) == bytes(hap)
    assert r["result"][0]["polish_rounds"] >= 2
    # ops are re-threaded consistently with the new consensus
    for i, rd in enumerate(reads):
        o = r["ops_out"][int(r["ops_out_off"][i]):int(r["ops_out_off"][i + 1])]
        assert (o != 2).sum() == len(cons) and (o != 3).sum() == len(rd)
