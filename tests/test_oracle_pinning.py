"""Pins the CPU oracle: the reference's own unit tests for this path, the runtime invariants
it asserts, the public known-answer vectors of the RNGs it uses, and the shared exp/log against libm."""
import ctypes as C
import math

import numpy as np

import oracle_ffi as O


def rng_from_state(s):
    r = O.Rng()
    for i, v in enumerate(s):
        r.s[i] = v
    return r


def test_xoshiro256starstar_kat(oracle):
    # published xoshiro256** vector for state {1,2,3,4} (also rand_xoshiro's own test vector)
    L = oracle.lib()
    r = rng_from_state([1, 2, 3, 4])
    got = [L.jo_rng_next_u64(C.byref(r)) for _ in range(10)]
    assert got == [11520, 0, 1509978240, 1215971899390074240, 1216172134540287360, 607988272756665600,
                   16172922978634559625, 8476171486693032832, 10595114339597558777, 2904607092377533576]


def test_splitmix64_kat(oracle):
    L = oracle.lib()
    x = C.c_uint64(1234567)
    got = [L.jo_splitmix64_next(C.byref(x)) for _ in range(5)]
    assert got == [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431,
                   16408922859458223821]
    r = O.Rng()
    L.jo_rng_seed_from_u64(C.byref(r), 0)
    assert list(r.s) == [0xe220a8397b1dcdaf, 0x6e789e6aa1b965f4, 0x06c45d188009454f, 0xf88bb8a8724c81ec]


def test_next_u32_is_upper_half(oracle):
    L = oracle.lib()
    a, b = rng_from_state([1, 2, 3, 4]), rng_from_state([1, 2, 3, 4])
    for _ in range(20):
        assert L.jo_rng_next_u32(C.byref(a)) == L.jo_rng_next_u64(C.byref(b)) >> 32


def test_gen_range_matches_widening_multiply_spec(oracle):
    """gen_range(0..n) = hi(v*n) for the first v whose lo(v*n) <= (n << lzcnt(n)) - 1 (rand 0.8.5)."""
    L = oracle.lib()
    for n in (1, 2, 3, 7, 60, 160, 1000003, (1 << 63) + 5):
        a, b = O.Rng(), O.Rng()
        L.jo_rng_seed_from_u64(C.byref(a), 99 + n % 1000)
        L.jo_rng_seed_from_u64(C.byref(b), 99 + n % 1000)
        zone = ((n << (64 - n.bit_length())) - 1) & (2 ** 64 - 1)
        for _ in range(200):
            got = L.jo_gen_range_usize(C.byref(a), n)
            while True:
                v = L.jo_rng_next_u64(C.byref(b))
                m = v * n
                if (m & (2 ** 64 - 1)) <= zone:
                    break
            assert got == m >> 64
            assert a.draws == b.draws


def test_gen_index_u32_path_and_range1_rejection(oracle):
    """gen_index(1) (u32 path) rejects every draw whose upper half has its top bit set."""
    L = oracle.lib()
    a, b = O.Rng(), O.Rng()
    L.jo_rng_seed_from_u64(C.byref(a), 5)
    L.jo_rng_seed_from_u64(C.byref(b), 5)
    for _ in range(300):
        assert L.jo_gen_index(C.byref(a), 1) == 0
        while L.jo_rng_next_u32(C.byref(b)) > 0x7FFFFFFF:
            pass
        assert a.draws == b.draws


def test_gen_bool_semantics(oracle):
    L = oracle.lib()
    r = O.Rng()
    L.jo_rng_seed_from_u64(C.byref(r), 1)
    d0 = r.draws
    assert L.jo_gen_bool(C.byref(r), 1.0) == 1 and r.draws == d0      # p == 1 never draws
    assert L.jo_gen_bool(C.byref(r), 0.0) == 0 and r.draws == d0 + 1  # p == 0 draws and is false
    a, b = O.Rng(), O.Rng()
    L.jo_rng_seed_from_u64(C.byref(a), 3)
    L.jo_rng_seed_from_u64(C.byref(b), 3)
    for p in np.linspace(0.001, 0.999, 97):
        want = L.jo_rng_next_u64(C.byref(b)) < int(p * 2.0 ** 64)
        assert bool(L.jo_gen_bool(C.byref(a), float(p))) == want


def test_choose_other_reservoir(oracle):
    L = oracle.lib()
    a, b = O.Rng(), O.Rng()
    L.jo_rng_seed_from_u64(C.byref(a), 11)
    L.jo_rng_seed_from_u64(C.byref(b), 11)
    for k in (2, 3, 4, 7):
        for old in range(k):
            for _ in range(50):
                got = L.jo_choose_other(C.byref(a), k, old)
                res, consumed = None, 0
                for c in range(k):
                    if c == old:
                        continue
                    consumed += 1
                    if L.jo_gen_index(C.byref(b), consumed) == 0:
                        res = c
                assert got == res and got != old and a.draws == b.draws


def test_choose_weighted(oracle):
    L = oracle.lib()
    r = O.Rng()
    L.jo_rng_seed_from_u64(C.byref(r), 42)
    w = np.array([0.0, 1.0, 0.0, 3.0], dtype=np.float64)
    counts = np.zeros(4)
    for _ in range(4000):
        counts[L.jo_choose_weighted(C.byref(r), O.f64p(w), 4)] += 1
    assert counts[0] == 0 and counts[2] == 0 and abs(counts[3] / counts[1] - 3) < 0.4
    z = np.zeros(5)
    assert L.jo_choose_weighted(C.byref(r), O.f64p(z), 5) == -1   # WeightedError::AllWeightsZero


# ---- the reference's own unit tests on this path -------------------------------------------------

def test_ref_cosine_similarity_test(oracle):
    """pseudo_mcmc.rs:876-897"""
    L = oracle.lib()

    def cos(rows):
        p = np.array(rows, dtype=np.float64)
        return L.jo_cosine_similarity(O.f64p(p), p.shape[0], p.shape[1], 0, 1)

    assert abs(1 - cos([[1, 1], [2, 2]])) < 1e-4
    assert abs(-1 - cos([[1, -3], [1, -3]])) < 1e-5
    assert abs(cos([[1, 3], [1, -3]])) < 1e-5
    assert math.sqrt(0.5) - abs(cos([[0, 100], [1, 100]])) < 1e-5


def test_ref_homop_length_test(oracle):
    """pseudo_mcmc.rs:899-904"""
    L = oracle.lib()
    xs = O.seq(b"ACCCCGTTTGGTT")
    out = np.zeros(len(xs), dtype=np.uintp)
    L.jo_homopolymer_length(O.u8p(xs), len(xs), O.szp(out))
    assert out.tolist() == [1, 4, 4, 4, 4, 1, 3, 3, 3, 2, 2, 2, 2]


def test_ref_reorder_test(oracle):
    """normalize.rs:68-74"""
    L = oracle.lib()
    arr = np.array([50, 40, 70, 60, 90], dtype=np.int64)
    idx = np.array([3, 0, 4, 1, 2], dtype=np.uint64)
    L.jo_reorder_i64(arr.ctypes.data_as(C.POINTER(C.c_int64)), O.u64p(idx), 5)
    assert arr.tolist() == [40, 60, 90, 50, 70]


def test_ref_rand_index_test(oracle):
    """misc.rs:461-465"""
    L = oracle.lib()
    pred = np.array([0, 0, 0, 1, 1, 1], dtype=np.uintp)
    answ = np.array([0, 0, 1, 1, 2, 2], dtype=np.uintp)
    assert abs(0.6666 - L.jo_rand_index(O.szp(pred), O.szp(answ), 6)) < 1e-4


# ---- shared math ------------------------------------------------------------------------------------

def test_jtk_exp_log_within_one_ulp_of_libm(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-745, 709, 20000), rng.uniform(-50, 5, 20000), [0.0, -0.0, 1e-300, -800.0]])
    for x in xs:
        a, b = L.jo_exp(float(x)), math.exp(x) if x < 709.7 else float("inf")
        if b == 0 or math.isinf(b):
            continue
        assert abs(a - b) <= abs(math.ulp(b)), x
    ys = np.concatenate([np.exp(rng.uniform(-700, 700, 20000)), rng.uniform(0.5, 2, 20000), [1.0, 5e-324]])
    for y in ys:
        a, b = L.jo_log(float(y)), math.log(y)
        assert abs(a - b) <= abs(math.ulp(b)) or (b == 0 and a == 0), y


def test_logsumexp_and_pvalues(oracle):
    L = oracle.lib()
    xs = np.array([-1.0, -2.0, -3.5], dtype=np.float64)
    assert abs(L.jo_logsumexp(O.f64p(xs), 3) - math.log(sum(math.exp(v) for v in xs))) < 1e-12
    n, p = 30, 0.04
    out = np.zeros(n + 1)
    L.jo_pvalues(p, n, O.f64p(out))
    from math import comb
    pmf = [comb(n, k) * p ** k * (1 - p) ** (n - k) for k in range(n + 1)]
    for i in range(n + 1):
        assert abs(out[i] - sum(pmf[i:])) < 1e-9   # i -> P(i <= X)
    assert abs(out[0] - 1.0) < 1e-12
