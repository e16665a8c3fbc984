/* rng.c -- CPU ORACLE (test infrastructure). Bit-exact restatement of the random sampling the reference
 * path performs through rand 0.8.5 / rand_core 0.6.4 / rand_xoshiro 0.6.0 (Cargo.lock:621-662).  Those
 * crates are NOT under /root/reference; the algorithms are restated from their published sources and
 * pinned by the public xoshiro256** / SplitMix64 known-answer vectors (tests/test_oracle_rng.py).
 * Reference call sites: local_clustering/mod.rs:97; pseudo_mcmc.rs:730,732,736; misc.rs:239-240,322,335;
 * likelihood_gains.rs:214-219,232,241,269.
 */
#include "jtk_oracle.h"

#include <stdlib.h>
#include <string.h>

/* rand_xoshiro::SplitMix64::next_u64 */
uint64_t jo_splitmix64_next(uint64_t *x) {
    *x += 0x9e3779b97f4a7c15ULL;
    uint64_t z = *x;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

/* Xoshiro256StarStar::seed_from_u64: state = four successive SplitMix64 outputs (from_rng over
 * SplitMix64::seed_from_u64(seed), little-endian fill). */
void jo_rng_seed_from_u64(jo_rng_t *rng, uint64_t seed) {
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) rng->s[i] = jo_splitmix64_next(&x);
    rng->draws = 0;
    rng->kind = 0;
}

/* rand_xoshiro 0.6.0 Xoroshiro128PlusPlus (phmm_likelihood_correction.rs:295-296): seed_from_u64 fills the two state words
 * from SplitMix64 like every generator of the crate; next_u64 = rotl(s0 + s1, 17) + s0, then s1 ^= s0,
 * s0 = rotl(s0, 49) ^ s1 ^ (s1 << 21), s1 = rotl(s1, 28) -- pinned by the reference implementation's known-answer vector for
 * state (1, 2) (tests/test_oracle_correction.py); next_u32 = the LOW half of next_u64 (recalled from the crate's source,
 * which is not under /root/reference: that choice is unpinned). */
void jo_rng128pp_seed_from_u64(jo_rng_t *rng, uint64_t seed) {
    uint64_t x = seed;
    rng->s[0] = jo_splitmix64_next(&x);
    rng->s[1] = jo_splitmix64_next(&x);
    rng->s[2] = rng->s[3] = 0;
    rng->draws = 0;
    rng->kind = 1;
}

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* Xoshiro256StarStar::next_u64 */
uint64_t jo_rng_next_u64(jo_rng_t *rng) {
    uint64_t *s = rng->s;
    if (rng->kind == 1) {
        const uint64_t s0 = s[0];
        uint64_t s1 = s[1];
        const uint64_t r = rotl64(s0 + s1, 17) + s0;
        s1 ^= s0;
        s[0] = rotl64(s0, 49) ^ s1 ^ (s1 << 21);
        s[1] = rotl64(s1, 28);
        rng->draws++;
        return r;
    }
    uint64_t result = rotl64(s[1] * 5, 7) * 9;
    uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    rng->draws++;
    return result;
}

/* Xoshiro256StarStar::next_u32: the upper half of next_u64 */
uint32_t jo_rng_next_u32(jo_rng_t *rng) {
    if (rng->kind == 1) return (uint32_t)jo_rng_next_u64(rng);
    return (uint32_t)(jo_rng_next_u64(rng) >> 32);
}

/* UniformInt<usize>::sample_single_inclusive(0, n-1) on a 64-bit target (rand 0.8.5
 * distributions/uniform.rs): zone = (range << lzcnt(range)) - 1; widening multiply rejection. */
uint64_t jo_gen_range_usize(jo_rng_t *rng, uint64_t n) {
    uint64_t range = n; /* callers guarantee n >= 1 (the reference asserts low < high) */
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
        uint64_t v = jo_rng_next_u64(rng);
        unsigned __int128 m = (unsigned __int128)v * range;
        uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
        if (lo <= zone) return hi;
    }
}

/* UniformInt<u32>::sample_single_inclusive(0, n-1): same in 32 bits on next_u32 */
uint32_t jo_gen_range_u32(jo_rng_t *rng, uint32_t n) {
    uint32_t range = n;
    uint32_t zone = (range << __builtin_clz(range)) - 1;
    for (;;) {
        uint32_t v = jo_rng_next_u32(rng);
        uint64_t m = (uint64_t)v * range;
        uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
        if (lo <= zone) return hi;
    }
}

/* rand::seq::gen_index: u32 sampling whenever the bound fits */
uint64_t jo_gen_index(jo_rng_t *rng, uint64_t ubound) {
    if (ubound <= 0xffffffffULL) return jo_gen_range_u32(rng, (uint32_t)ubound);
    return jo_gen_range_usize(rng, ubound);
}

/* Rng::gen_bool -> Bernoulli::new(p).sample: p == 1 never draws; p_int = (p * 2^64) as u64 */
int jo_gen_bool(jo_rng_t *rng, double p) {
    if (p == 1.0) return 1;
    double scaled = p * 18446744073709551616.0; /* 2.0 * (1u64 << 63) as f64 */
    uint64_t p_int;
    if (!(scaled > 0.0))
        p_int = 0;
    else if (scaled >= 18446744073709551616.0)
        p_int = UINT64_MAX; /* Rust `as u64` saturates; unreachable for p < 1 */
    else
        p_int = (uint64_t)scaled;
    uint64_t v = jo_rng_next_u64(rng);
    return v < p_int;
}

/* IteratorRandom::choose on `(0..k).filter(|&c| c != old)`.  Filter's size_hint lower bound is 0, so
 * rand's choose takes its one-at-a-time reservoir branch: the i-th yielded element (i = 1, 2, ...)
 * replaces the result iff gen_index(i) == 0. */
uint64_t jo_choose_other(jo_rng_t *rng, uint64_t k, uint64_t old) {
    uint64_t result = (uint64_t)-1, consumed = 0;
    for (uint64_t c = 0; c < k; c++) {
        if (c == old) continue;
        consumed++;
        if (jo_gen_index(rng, consumed) == 0) result = c;
    }
    return result;
}

/* SliceRandom::choose_weighted -> WeightedIndex<f64>::new + sample (rand 0.8.5
 * distributions/weighted_index.rs, uniform.rs UniformFloat<f64>). */
int64_t jo_choose_weighted(jo_rng_t *rng, const double *w, size_t n) {
    if (n == 0) return -1; /* WeightedError::NoItem */
    double total = w[0];
    if (!(total >= 0.0)) return -1; /* InvalidWeight */
    /* cumulative_weights has n-1 entries: running total BEFORE adding w[i], i = 1..n-1 */
    double cum_static[256];
    double *cum = cum_static;
    double *heap = 0;
    if (n - 1 > 256) {
        heap = (double *)malloc((n - 1) * sizeof(double));
        cum = heap;
    }
    for (size_t i = 1; i < n; i++) {
        if (!(w[i] >= 0.0)) {
            if (heap) free(heap);
            return -1;
        }
        cum[i - 1] = total;
        total += w[i];
    }
    if (total == 0.0) { /* AllWeightsZero */
        if (heap) free(heap);
        return -1;
    }
    /* UniformFloat::new(0, total): scale = total, decreased by one ulp while scale*max_rand >= high */
    double scale = total;
    const double max_rand = 1.0 - 0x1p-52; /* (u64::MAX >> 12).into_float_with_exponent(0) - 1.0 */
    for (;;) {
        if (!(scale * max_rand + 0.0 >= total)) break;
        uint64_t b;
        memcpy(&b, &scale, 8);
        b -= 1;
        memcpy(&scale, &b, 8);
    }
    uint64_t bits = (jo_rng_next_u64(rng) >> 12) | 0x3ff0000000000000ULL;
    double value1_2;
    memcpy(&value1_2, &bits, 8);
    double chosen = (value1_2 - 1.0) * scale + 0.0;
    /* first index whose cumulative weight is > chosen (partition point of `w <= chosen`) */
    size_t lo = 0, hi = n - 1;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (cum[mid] <= chosen)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (heap) {
        free(heap);
    }
    return (int64_t)lo;
}
