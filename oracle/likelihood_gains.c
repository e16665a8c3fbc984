/* likelihood_gains.c -- CPU ORACLE (test infrastructure). Restatement of haplotyper/src/likelihood_gains.rs.
 * The table/p-value code follows the Rust exactly.  estimate_gain's simulation calls kiley
 * (generate_seq, Generate::gen, likelihood_antidiagonal_bootstrap), which is absent: those three are the
 * OWN-SPEC versions of phmm.c, so the calibrated numbers are this build's, not kiley's (parity unpinned).
 */
#include <stdlib.h>
#include <string.h>

#include "jtk_math.h"
#include "jtk_oracle.h"

/* likelihood_gains.rs:79-87 */
double jo_gains_expected(const jtk_gains_t *g, size_t homop_len, int diff_type) {
    /* assert!(0 < homop_len): callers on the path never pass 0 for positions inside the template */
    if (homop_len == 0) homop_len = 1;
    size_t h = homop_len < g->max_homopolymer_len ? homop_len : g->max_homopolymer_len;
    switch (diff_type) {
        case JTK_DIFF_SUBST: return g->subst[h - 1].gain;
        case JTK_DIFF_DEL: return g->deletions[h - 1].gain;
        default: return g->insertions[h - 1].gain;
    }
}

/* likelihood_gains.rs:131-137 */
static double logsumexp2(double x, double y) {
    if (y < x) return x + jtk_log(1.0 + jtk_exp(y - x));
    return y + jtk_log(1.0 + jtk_exp(x - y));
}

/* likelihood_gains.rs:115-129: i -> P(i <= X | n, prob) */
void jo_pvalues(double prob, size_t n, double *out) {
    double ln = jtk_log(prob), in_ln = jtk_log(1.0 - prob);
    out[0] = in_ln * (double)n;
    for (size_t k = 0; k < n; k++) {
        double offset = ln + jtk_log((double)(n - k)) - in_ln - jtk_log((double)(k + 1));
        out[k + 1] = out[k] + offset;
    }
    for (size_t k = n; k-- > 0;) out[k] = logsumexp2(out[k + 1], out[k]);
    for (size_t k = 0; k <= n; k++) out[k] = jtk_exp(out[k]);
}

static int cmp_f64(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}
/* select_nth_unstable_by(k).1 == k-th smallest */
static double nth(double *xs, size_t n, size_t k) {
    qsort(xs, n, sizeof(double), cmp_f64);
    return xs[k];
}

static const uint8_t BASES[4] = {'A', 'C', 'G', 'T'};

/* likelihood_gains.rs:213-222 */
static void sample_triple(jo_rng_t *rng, uint8_t *right, uint8_t *homop, uint8_t *left) {
    double w[4];
    *homop = BASES[jo_gen_index(rng, 4)];
    for (int i = 0; i < 4; i++) w[i] = BASES[i] != *homop ? 1.0 : 0.0;
    *right = BASES[jo_choose_weighted(rng, w, 4)];
    for (int i = 0; i < 4; i++) w[i] = (BASES[i] != *homop && BASES[i] != *right) ? 1.0 : 0.0;
    *left = BASES[jo_choose_weighted(rng, w, 4)];
}

/* likelihood_gains.rs:224-251; hap buffers need len + 3 */
static void gen_diff_haplotypes(jo_rng_t *rng, size_t len, int diff_type, uint8_t *hap1, size_t *l1,
                                uint8_t *hap2, size_t *l2) {
    uint8_t right, center, left;
    double w[4];
    sample_triple(rng, &right, &center, &left);
    uint8_t c2[32];
    size_t n2 = len;
    for (size_t i = 0; i < len; i++) c2[i] = center;
    if (diff_type == JTK_DIFF_SUBST) {
        for (int i = 0; i < 4; i++) w[i] = BASES[i] != center ? 1.0 : 0.0;
        c2[0] = BASES[jo_choose_weighted(rng, w, 4)];
    } else if (diff_type == JTK_DIFF_DEL) {
        memmove(c2, c2 + 1, len - 1);
        n2 = len - 1;
    } else {
        for (int i = 0; i < 4; i++) w[i] = BASES[i] != center ? 1.0 : 0.0;
        uint8_t diff = BASES[jo_choose_weighted(rng, w, 4)];
        /* Vec::insert(1, diff): for len == 1 this appends */
        size_t at = 1 <= len ? 1 : len;
        memmove(c2 + at + 1, c2 + at, len - at);
        c2[at] = diff;
        n2 = len + 1;
    }
    size_t p = 0;
    hap1[p++] = right;
    for (size_t i = 0; i < len; i++) hap1[p++] = center;
    hap1[p++] = left;
    *l1 = p;
    p = 0;
    hap2[p++] = right;
    for (size_t i = 0; i < n2; i++) hap2[p++] = c2[i];
    hap2[p++] = left;
    *l2 = p;
}

/* likelihood_gains.rs:253-315 (rayon par_iter over i is order-independent: each i has its own RNG) */
static jtk_gain_profile_t gain_of(const jtk_hmm_t *fwd, const jtk_hmm_t *rev, uint64_t seed, size_t seq_len,
                                  size_t band, size_t len, int diff_type) {
    enum { SAMPLE_NUM = 100, GAIN_POS = SAMPLE_NUM / 10, PROB_POS = SAMPLE_NUM * 2 / 3, SEQ_NUM = 50 };
    double medians[SAMPLE_NUM], probs[SAMPLE_NUM];
    size_t half = seq_len / 2;
    size_t cap = 2 * half + 64;
    uint8_t *tmpl = (uint8_t *)malloc(cap), *diff = (uint8_t *)malloc(cap);
    uint8_t *seg1 = (uint8_t *)malloc(half + 1), *seg2 = (uint8_t *)malloc(half + 1);
    size_t rcap = 3 * cap + 16;
    uint8_t *read = (uint8_t *)malloc(rcap);
    for (int i = 0; i < SAMPLE_NUM; i++) {
        jo_rng_t rng;
        jo_rng_seed_from_u64(&rng, (uint64_t)i + seed);
        jo_generate_seq(&rng, half, seg1);
        jo_generate_seq(&rng, half, seg2);
        uint8_t h1[40], h2[40];
        size_t l1, l2;
        gen_diff_haplotypes(&rng, len, diff_type, h1, &l1, h2, &l2);
        size_t tl = 0, dl = 0;
        memcpy(tmpl, seg1, half);
        tl = half;
        memcpy(tmpl + tl, h1, l1);
        tl += l1;
        memcpy(tmpl + tl, seg2, half);
        tl += half;
        memcpy(diff, seg1, half);
        dl = half;
        memcpy(diff + dl, h2, l2);
        dl += l2;
        memcpy(diff + dl, seg2, half);
        dl += half;
        double lk_diff[SEQ_NUM];
        for (int t = 0; t < SEQ_NUM; t++) {
            const jtk_hmm_t *h = (t % 2 == 0) ? fwd : rev;
            size_t rl = jo_phmm_gen(h, diff, dl, &rng, read, rcap);
            double lk_base = jo_phmm_likelihood_bootstrap(h, tmpl, tl, read, rl, band);
            double lk_d = jo_phmm_likelihood_bootstrap(h, diff, dl, read, rl, band);
            lk_diff[t] = lk_d - lk_base;
        }
        double expected_gain = nth(lk_diff, SEQ_NUM, SEQ_NUM / 2);
        double min_gain = diff_type == JTK_DIFF_SUBST ? expected_gain / 10.0 : 0.0001;
        size_t null_cnt = 0;
        for (int t = 0; t < SEQ_NUM; t++) {
            const jtk_hmm_t *h = (t % 2 == 0) ? fwd : rev;
            size_t rl = jo_phmm_gen(h, tmpl, tl, &rng, read, rcap);
            double lk_base = jo_phmm_likelihood_bootstrap(h, tmpl, tl, read, rl, band);
            double lk_d = jo_phmm_likelihood_bootstrap(h, diff, dl, read, rl, band);
            if (lk_base + min_gain < lk_d) null_cnt++;
        }
        medians[i] = expected_gain;
        probs[i] = (double)null_cnt / (double)SEQ_NUM;
    }
    jtk_gain_profile_t gp;
    gp.gain = nth(medians, SAMPLE_NUM, GAIN_POS);
    gp.prob = nth(probs, SAMPLE_NUM, PROB_POS);
    if (gp.prob < 0.000000001) gp.prob = 0.000000001; /* prob.max(1e-9) */
    free(tmpl);
    free(diff);
    free(seg1);
    free(seg2);
    free(read);
    return gp;
}

/* likelihood_gains.rs:162-184 */
void jo_estimate_gain(const jtk_hmm_t *fwd, const jtk_hmm_t *rev, uint64_t seed, size_t seq_len,
                      size_t band, size_t homop_len, jtk_gains_t *out) {
    memset(out, 0, sizeof *out);
    if (homop_len > JTK_GAINS_MAX_HOMOP) homop_len = JTK_GAINS_MAX_HOMOP;
    out->max_homopolymer_len = (uint32_t)homop_len;
    for (size_t len = 1; len <= homop_len; len++)
        out->subst[len - 1] = gain_of(fwd, rev, seed, seq_len, band, len, JTK_DIFF_SUBST);
    for (size_t len = 1; len <= homop_len; len++)
        out->deletions[len - 1] = gain_of(fwd, rev, seed, seq_len, band, len, JTK_DIFF_DEL);
    for (size_t len = 1; len <= homop_len; len++)
        out->insertions[len - 1] = gain_of(fwd, rev, seed, seq_len, band, len, JTK_DIFF_INS);
}

/* likelihood_gains.rs:186-192 */
void jo_estimate_gain_default(const jtk_hmm_t *fwd, const jtk_hmm_t *rev, jtk_gains_t *out) {
    jo_estimate_gain(fwd, rev, 309423, 100, 10, 3, out);
}

/* kiley gen_seq::introduce_errors(seq, rng, 0, 1, 0) (likelihood_gains.rs:18): kiley is not under /root/reference; OWN
 * SPECIFICATION after the crate's published source as recalled: the operations [Match x (len - 1), Del x 1] are shuffled
 * with rand 0.8.5 SliceRandom::shuffle (for i in (1..len).rev(): swap(i, gen_index(i + 1))) and applied in order, so the
 * base at the Del's final position is dropped.  Parity with kiley unpinned. */
static size_t introduce_one_deletion(jo_rng_t *rng, const uint8_t *seq, size_t len, uint8_t *out) {
    size_t del_pos = len - 1;
    for (size_t i = len - 1; i >= 1; i--) {
        const size_t j = (size_t)jo_gen_index(rng, i + 1);
        if (del_pos == i)
            del_pos = j;
        else if (del_pos == j)
            del_pos = i;
    }
    size_t p = 0;
    for (size_t i = 0; i < len; i++)
        if (i != del_pos) out[p++] = seq[i];
    return p;
}

/* estimate_minimum_gain (likelihood_gains.rs:6-39; the reference's constants: seed 23908, 1000 samples x 500 reads of a
 * 100-base template, band 25, floor 1.0): per sample the median over reads of lk(read | hap1) - lk(read | hap1 minus one
 * base), reads drawn from hap1; the third smallest of the samples' medians, at least 1. */
double jo_estimate_minimum_gain(const jtk_hmm_t *fwd, const jtk_hmm_t *rev, uint64_t seed, size_t sample_num, size_t seq_num,
                                size_t len, size_t band, int n_threads) {
    if (sample_num < 3 || seq_num == 0 || len < 2) return 0.0 / 0.0; /* medians[2] / lks[SEQ_NUM / 2] out of bounds */
    double *medians = (double *)malloc(sample_num * sizeof(double));
    (void)n_threads;
#pragma omp parallel for schedule(dynamic) num_threads(n_threads > 0 ? n_threads : 1)
    for (long s = 0; s < (long)sample_num; s++) {
        jo_rng_t rng;
        jo_rng_seed_from_u64(&rng, seed + (uint64_t)s);
        uint8_t *hap1 = (uint8_t *)malloc(len), *hap2 = (uint8_t *)malloc(len);
        const size_t rcap = 3 * len + 16;
        uint8_t *read = (uint8_t *)malloc(rcap);
        double *lks = (double *)malloc(seq_num * sizeof(double));
        jo_generate_seq(&rng, len, hap1);
        const size_t l2 = introduce_one_deletion(&rng, hap1, len, hap2);
        for (size_t t = 0; t < seq_num; t++) {
            const jtk_hmm_t *h = (t % 2 == 0) ? fwd : rev;
            const size_t rl = jo_phmm_gen(h, hap1, len, &rng, read, rcap);
            const double lk_base = jo_phmm_likelihood_bootstrap(h, hap1, len, read, rl, band);
            const double lk_diff = jo_phmm_likelihood_bootstrap(h, hap2, l2, read, rl, band);
            lks[t] = lk_base - lk_diff;
        }
        medians[s] = nth(lks, seq_num, seq_num / 2);
        free(hap1);
        free(hap2);
        free(read);
        free(lks);
    }
    qsort(medians, sample_num, sizeof(double), cmp_f64);
    const double m = medians[2];
    free(medians);
    return m > 1.0 ? m : 1.0; /* .max(MIN_REQ) */
}
