/* pseudo_mcmc.c -- CPU ORACLE (test infrastructure).  Line-by-line restatement of
 * haplotyper/src/local_clustering/pseudo_mcmc.rs.  Every function names the Rust lines it follows.
 * Rust iterator semantics that decide results are kept: max_by returns the LAST maximum, min_by the
 * FIRST minimum, f64 sums run left to right from 0.0, f64::max ignores NaN.
 */
#include <stdlib.h>
#include <string.h>

#include "jtk_math.h"
#include "jtk_oracle.h"

#define NUM_ROW JTK_NUM_ROW
#define COPY_SIZE JTK_COPY_SIZE

/* ---- the reference's trace! rows (log level Trace) of one chunk's clustering, into a text buffer.  The sink belongs to the THREAD
 * that set it: the caller (tests/oracle_ffi.py: trace_chunk) runs ONE chunk on one thread (jo_cluster_chunks with n_threads = 1
 * stays on the calling thread) while it is set; the worker threads of any other call see no sink and write nothing.  Rows kept: TOTAL :467, CAND :471, PICK :539,
 * DUMP :126, RANGE :236, LK :250,:256, COUNTS :262.  (Not kept: the per-column PVALUE / RAWCOUNT / FILTER rows, REMOVE, VARS.)
 * Rust's {:.N} and C's %.Nf both print the correctly rounded decimal; NaN is "NaN" in Rust. */
#include <stdarg.h>
#include <stdio.h>
static __thread jo_trace_t *jo_trace_sink = NULL;
void jo_trace_set(jo_trace_t *t) { jo_trace_sink = t; }
static void trace_row(const char *fmt, ...) {
    jo_trace_t *t = jo_trace_sink;
    if (!t) return;
    char line[512];
    va_list ap;
    va_start(ap, fmt);
    int m = vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    if (m < 0) return;
    if ((size_t)m >= sizeof line) m = (int)sizeof line - 1;
    if (t->len + (size_t)m + 1 <= t->cap) {
        memcpy(t->text + t->len, line, (size_t)m);
        t->text[t->len + (size_t)m] = '\n';
    }
    t->len += (size_t)m + 1; /* keeps counting beyond cap: the caller learns the size it needs */
}
/* {x:.N} of an f64 */
static const char *fx(char *buf, size_t cap, double x, int prec) {
    if (x != x)
        snprintf(buf, cap, "NaN");
    else
        snprintf(buf, cap, "%.*f", prec, x);
    return buf;
}

/* pseudo_mcmc.rs:168-178 */
static void pos_to_bp_and_difftype(size_t pos, size_t *bp, int *dt) {
    size_t op = pos % NUM_ROW;
    *bp = pos / NUM_ROW;
    if (op < 4)
        *dt = JTK_DIFF_SUBST;
    else if (op < 8 + COPY_SIZE)
        *dt = JTK_DIFF_INS;
    else
        *dt = JTK_DIFF_DEL;
}

/* pseudo_mcmc.rs:195-211 */
void jo_homopolymer_length(const uint8_t *xs, size_t n, size_t *out) {
    if (n == 0) return;
    uint8_t current = xs[0];
    size_t length = 0, w = 0;
    for (size_t i = 0; i < n; i++) {
        if (xs[i] == current) {
            length++;
        } else {
            for (size_t t = 0; t < length; t++) out[w++] = length;
            current = xs[i];
            length = 1;
        }
    }
    for (size_t t = 0; t < length; t++) out[w++] = length;
}

/* pseudo_mcmc.rs:141-165 */
void jo_compress_small_gains(double *profiles, size_t n, size_t cols, const uint8_t *tmpl, size_t tl,
                             const jtk_gains_t *gains) {
    const double MIN_REQ_FRACTION = 0.5;
    if (n == 0) return;
    size_t *homop = (size_t *)malloc((tl ? tl : 1) * sizeof(size_t));
    jo_homopolymer_length(tmpl, tl, homop);
    double *min_req = (double *)malloc(cols * sizeof(double));
    for (size_t pos = 0; pos < cols; pos++) {
        size_t bp;
        int dt;
        pos_to_bp_and_difftype(pos, &bp, &dt);
        size_t homop_len = (bp < tl) ? homop[bp] : 1; /* .get(bp).unwrap_or(&1) */
        min_req[pos] = jo_gains_expected(gains, homop_len, dt) * MIN_REQ_FRACTION;
    }
    for (size_t r = 0; r < n; r++) {
        double *prof = profiles + r * cols;
        for (size_t pos = 0; pos < cols; pos++) {
            double x = prof[pos];
            double ax = x < 0 ? -x : x;
            if (ax < min_req[pos]) prof[pos] = 0.0;
        }
    }
    free(homop);
    free(min_req);
}

/* pseudo_mcmc.rs:602-615 */
double jo_cosine_similarity(const double *profiles, size_t n, size_t cols, size_t i, size_t j) {
    double ip = 0.0, isumsq = 0.0, jsumsq = 0.0;
    for (size_t r = 0; r < n; r++) {
        double x = profiles[r * cols + i], y = profiles[r * cols + j];
        double ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
        if (JO_POS_THR < ax && JO_POS_THR < ay) {
            ip = ip + x * y;
            isumsq = isumsq + x * x;
            jsumsq = jsumsq + y * y;
        }
    }
    if (isumsq == 0.0) return 0.0;
    return ip / __builtin_sqrt(isumsq) / __builtin_sqrt(jsumsq);
}

/* pseudo_mcmc.rs:618-633 */
double jo_sokal_michener(const double *profiles, size_t n, size_t cols, size_t i, size_t j) {
    size_t mat = 0, mism = 0;
    for (size_t r = 0; r < n; r++) {
        double x = profiles[r * cols + i], y = profiles[r * cols + j];
        double ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
        if (JO_POS_THR < ax && JO_POS_THR < ay) {
            if (0.0 < x * y)
                mat++;
            else
                mism++;
        }
    }
    size_t total = mat + mism;
    if (total == 0) return 0.0;
    return (double)(mism > mat ? mism : mat) / (double)total;
}

/* pseudo_mcmc.rs:636-638 */
double jo_poisson_lk(size_t x, double lambda) {
    double s = 0.0;
    for (size_t c = 1; c < x + 1; c++) s += jtk_log((double)c);
    return (double)x * jtk_log(lambda) - lambda - s;
}

/* pseudo_mcmc.rs:641-645 */
double jo_max_poisson_lk(size_t x, double lambda, size_t c_start, size_t c_end) {
    double m = -__builtin_inf();
    for (size_t c = (c_start > 1 ? c_start : 1); c <= c_end; c++)
        m = jtk_fmax(m, jo_poisson_lk(x, lambda * (double)c));
    return m;
}

/* pseudo_mcmc.rs:314-339; profiles column `pos`, strands[r] != 0 == forward */
static int is_explainable_by_strandedness(const double *profiles, size_t n, size_t cols, size_t pos,
                                          const uint8_t *strands) {
    size_t strand_count[2] = {0, 0}, sign_count[2] = {0, 0}, obs_count[2][2] = {{0, 0}, {0, 0}};
    for (size_t r = 0; r < n; r++) {
        double lk = profiles[r * cols + pos];
        double a = lk < 0 ? -lk : lk;
        if (!(a > 0.0001)) continue;
        size_t s = strands[r] ? 1 : 0;
        size_t g = (jtk_f64_bits(lk) >> 63) ? 0 : 1; /* is_sign_positive */
        strand_count[s]++;
        sign_count[g]++;
        obs_count[s][g]++;
    }
    size_t sum = strand_count[0] + strand_count[1];
    if (sum == 0) return 0;
    double chisq = 0.0;
    for (size_t s = 0; s < 2; s++) {
        double inner = 0.0;
        for (size_t g = 0; g < 2; g++) {
            double expected = (double)(strand_count[s] * sign_count[g]) / (double)sum;
            double d = (double)obs_count[s][g] - expected;
            inner += d * d / expected; /* 0/0 = NaN when a strand or a sign is absent: kept */
        }
        chisq += inner;
    }
    return chisq < 10.0;
}

/* pseudo_mcmc.rs:497-514 */
static int is_in_short_homopolymer(size_t pos, const size_t *homop, const uint8_t *tmpl, size_t tl) {
    size_t x;
    int dt;
    pos_to_bp_and_difftype(pos, &x, &dt);
    if (dt == JTK_DIFF_INS) {
        size_t bi = pos % NUM_ROW - 4;
        uint8_t base = bi < 4 ? (uint8_t)"ACGT"[bi] : 0;
        /* the reference indexes template[x-1] / template[x] unconditionally; callers mask 7 bases at
         * both ends first (pseudo_mcmc.rs:443-446), so 1 <= x < tl here */
        size_t prev_len = (0 < x) ? homop[x - 1] + (tmpl[x - 1] == base) : 0;
        size_t next_len = (x < tl) ? homop[x] + (tmpl[x] == base) : 0;
        return prev_len <= JO_MAX_HOMOP_LENGTH && next_len <= JO_MAX_HOMOP_LENGTH;
    }
    if (dt == JTK_DIFF_DEL && x < tl) return homop[x] <= JO_MAX_HOMOP_LENGTH;
    return 1;
}

typedef struct pvalues {
    size_t max_homop, total;
    double *tab[3][JTK_GAINS_MAX_HOMOP]; /* [diff_type][homop-1][count] */
} pvalues_t;

/* likelihood_gains.rs:88-111 */
static void pvalues_new(const jtk_gains_t *g, size_t total, pvalues_t *p) {
    p->max_homop = g->max_homopolymer_len;
    p->total = total;
    for (size_t h = 0; h < p->max_homop; h++) {
        const jtk_gain_profile_t *src[3] = {&g->subst[h], &g->deletions[h], &g->insertions[h]};
        for (int t = 0; t < 3; t++) {
            p->tab[t][h] = (double *)malloc((total + 1) * sizeof(double));
            jo_pvalues(src[t]->prob, total, p->tab[t][h]);
        }
    }
}
static void pvalues_free(pvalues_t *p) {
    for (size_t h = 0; h < p->max_homop; h++)
        for (int t = 0; t < 3; t++) free(p->tab[t][h]);
}
/* likelihood_gains.rs:149-158 */
static double pvalues_pvalue(const pvalues_t *p, size_t homop_len, int dt, size_t count) {
    size_t h = homop_len < p->max_homop ? homop_len : p->max_homop;
    return p->tab[dt][h - 1][count];
}

/* pseudo_mcmc.rs:476-495 */
static int has_small_pvalue(size_t pos, double gain, size_t count, const size_t *homop, size_t tl,
                            const pvalues_t *pv, const jtk_gains_t *gains, size_t template_len) {
    const double EXPT_GAIN_FACTOR = 0.8, PVALUE = 0.05;
    size_t bp;
    int dt;
    pos_to_bp_and_difftype(pos, &bp, &dt);
    size_t homop_len = bp < tl ? homop[bp] : 0;
    double pvalue = pvalues_pvalue(pv, homop_len, dt, count);
    double expt = jo_gains_expected(gains, homop_len, dt) * EXPT_GAIN_FACTOR;
    pvalue = (double)template_len * pvalue;
    return ((double)count * expt < gain) && (pvalue < PVALUE / (double)template_len);
}

/* pseudo_mcmc.rs:590-600: LAST maximum among flag == 0 */
static int64_t find_next_variants(const double *score, const uint8_t *sel, size_t n) {
    int64_t best = -1;
    double bv = 0;
    for (size_t i = 0; i < n; i++) {
        if (sel[i] != 0) continue;
        if (best < 0 || !(score[i] < bv)) {
            best = (int64_t)i;
            bv = score[i];
        }
    }
    return best;
}

/* pseudo_mcmc.rs:516-575 */
static size_t pick_filtered_profiles(const size_t *ppos, const double *pscore, size_t np,
                                     const double *profiles, size_t n, size_t cols, size_t cluster_num,
                                     size_t *pos_out, double *score_out) {
    const size_t ROUND = 3;
    uint8_t *sel = (uint8_t *)calloc(np ? np : 1, 1);
    size_t per_round = cluster_num > 2 ? cluster_num : 2;
    for (size_t round = 0; round < ROUND; round++) {
        for (size_t i = 0; i < np; i++)
            if (sel[i] == 3) sel[i] = 0;
        for (size_t it = 0; it < per_round; it++) {
            int64_t nx = find_next_variants(pscore, sel, np);
            if (nx < 0) break;
            size_t picked_pos = ppos[nx];
            size_t picked_bp = picked_pos / NUM_ROW;
            if (jo_trace_sink) { /* :538-539: position in bp, DiffType as S / I / D, lk */
                size_t bp_;
                int dt_;
                char b_[64];
                pos_to_bp_and_difftype(picked_pos, &bp_, &dt_);
                trace_row("PICK\t%zu\t%s\t%s", bp_, dt_ == JTK_DIFF_SUBST ? "S" : (dt_ == JTK_DIFF_DEL ? "D" : "I"),
                          fx(b_, sizeof b_, pscore[nx], 3));
            }
            sel[nx] = 1;
            for (size_t i = 0; i < np; i++) {
                if (!(sel[i] == 0 || sel[i] == 3)) continue;
                size_t bp = ppos[i] / NUM_ROW;
                size_t diff = (bp > picked_bp ? bp : picked_bp) - (bp < picked_bp ? bp : picked_bp);
                if (diff < JO_MASK_LENGTH) {
                    sel[i] = 2;
                } else {
                    double sok = jo_sokal_michener(profiles, n, cols, picked_pos, ppos[i]);
                    double cs = jo_cosine_similarity(profiles, n, cols, picked_pos, ppos[i]);
                    double acs = cs < 0 ? -cs : cs;
                    if (0.8 < sok || 0.8 < acs) sel[i] = 3;
                }
            }
        }
    }
    size_t nout = 0;
    for (size_t i = 0; i < np; i++)
        if (sel[i] == 1) {
            pos_out[nout] = ppos[i];
            score_out[nout] = pscore[i];
            nout++;
        }
    free(sel);
    return nout;
}

/* pseudo_mcmc.rs:426-474 (+ column_sum :577-588) */
size_t jo_filter_profiles(const uint8_t *tmpl, size_t tl, const double *profiles, size_t n,
                          const uint8_t *strands, const jo_cluster_config_t *cfg, size_t *pos_out,
                          double *score_out) {
    size_t cluster_num = cfg->copy_num;
    double coverage = cfg->coverage;
    const jtk_gains_t *gains = cfg->gains;
    size_t cols = NUM_ROW * (tl + 1);
    pvalues_t pv;
    pvalues_new(gains, n, &pv);
    size_t *homop = (size_t *)malloc((tl ? tl : 1) * sizeof(size_t));
    jo_homopolymer_length(tmpl, tl, homop);
    double *tot_gain = (double *)calloc(cols, sizeof(double));
    size_t *tot_cnt = (size_t *)calloc(cols, sizeof(size_t));
    for (size_t r = 0; r < n; r++)
        for (size_t pos = 0; pos < cols; pos++) {
            double g = profiles[r * cols + pos];
            if (JO_POS_THR < g) {
                tot_gain[pos] += g;
                tot_cnt[pos]++;
            }
        }
    size_t temp_len = cols / NUM_ROW;
    size_t *ppos = (size_t *)malloc(cols * sizeof(size_t));
    double *pscore = (double *)malloc(cols * sizeof(double));
    size_t np = 0;
    for (size_t pos = 0; pos < cols; pos++) {
        size_t bp = pos / NUM_ROW, row = pos % NUM_ROW;
        if (!(JO_MASK_LENGTH <= bp && bp + JO_MASK_LENGTH <= temp_len)) continue; /* bp <= temp_len - 7 */
        if (!(row < 8 || row == 8 + COPY_SIZE)) continue;
        if (!is_in_short_homopolymer(pos, homop, tmpl, tl)) continue;
        if (!has_small_pvalue(pos, tot_gain[pos], tot_cnt[pos], homop, tl, &pv, gains, temp_len)) continue;
        if (!is_explainable_by_strandedness(profiles, n, cols, pos, strands)) continue;
        double max_lk = 0;
        for (size_t k = 1; k < cluster_num + 1; k++) {
            double v = jo_poisson_lk(tot_cnt[pos], coverage * (double)k);
            if (k == 1 || !(v < max_lk)) max_lk = v;
        }
        double total_lk = max_lk + tot_gain[pos];
        if (0.0 < total_lk) {
            ppos[np] = pos;
            pscore[np] = total_lk;
            np++;
        }
    }
    if (jo_trace_sink) { /* :467-472 */
        trace_row("TOTAL\t%zu", np);
        for (size_t i = 0; i < np; i++) {
            char b_[64];
            trace_row("CAND\t%zu\t%zu\t%s\t%zu", ppos[i] / NUM_ROW, ppos[i] % NUM_ROW, fx(b_, sizeof b_, pscore[i], 1),
                      tot_cnt[ppos[i]]);
        }
    }
    size_t d = pick_filtered_profiles(ppos, pscore, np, profiles, n, cols, cluster_num, pos_out, score_out);
    free(ppos);
    free(pscore);
    free(tot_gain);
    free(tot_cnt);
    free(homop);
    pvalues_free(&pv);
    return d;
}

/* ---------------------------------------- LKCount (pseudo_mcmc.rs:797-845) ------------------------- */
typedef struct lkcount {
    double total_gain;
    size_t num_pos, num_neg, num_zero;
} lkcount_t;

static int lk_is_informative(const lkcount_t *c) {
    const double POS_FRAC = 0.70;
    double cov = (double)(c->num_pos + c->num_neg) + 0.0000001;
    return 0.0 < c->total_gain && POS_FRAC < (double)c->num_pos / cov;
}
static void lk_add(lkcount_t *c, double x) {
    c->total_gain += x;
    if (JO_POS_THR < x)
        c->num_pos++;
    else if (x < -JO_POS_THR)
        c->num_neg++;
    else
        c->num_zero++;
}
static void lk_sub(lkcount_t *c, double x) {
    c->total_gain -= x;
    if (JO_POS_THR < x)
        c->num_pos--;
    else if (x < -JO_POS_THR)
        c->num_neg--;
    else
        c->num_zero--;
}

/* pseudo_mcmc.rs:847-869; lks is k x dim */
static void get_used_columns(const lkcount_t *lks, size_t k, size_t dim, uint8_t *to_uses) {
    const double IN_POS_RATIO = 2.0;
    memset(to_uses, 0, dim);
    for (size_t c = 0; c < k; c++)
        for (size_t d = 0; d < dim; d++) to_uses[d] |= (uint8_t)lk_is_informative(&lks[c * dim + d]);
    for (size_t d = 0; d < dim; d++) {
        size_t pos_in_use = 0, pos_in_neg = 0;
        for (size_t c = 0; c < k; c++) {
            const lkcount_t *x = &lks[c * dim + d];
            if (0.0 < x->total_gain) pos_in_use += x->num_pos;
            if (x->total_gain <= 0.0) pos_in_neg += x->num_pos;
        }
        to_uses[d] &= (uint8_t)((double)pos_in_neg * IN_POS_RATIO < (double)pos_in_use);
    }
}

/* pseudo_mcmc.rs:785-795 */
static double get_lk(const lkcount_t *lks, const size_t *clusters, size_t k, size_t dim,
                     const double *size_to_lk, uint8_t *use_scratch) {
    get_used_columns(lks, k, dim, use_scratch);
    double lk = 0.0;
    for (size_t c = 0; c < k; c++) lk += size_to_lk[clusters[c]];
    for (size_t c = 0; c < k; c++)
        for (size_t d = 0; d < dim; d++)
            if (use_scratch[d]) lk += jtk_fmax(lks[c * dim + d].total_gain, 0.0);
    return lk;
}

/* pseudo_mcmc.rs:764-783 */
static void flip(const double *data, size_t dim, size_t *assign, size_t idx, size_t to, lkcount_t *lks,
                 size_t *clusters) {
    size_t from = assign[idx];
    clusters[from]--;
    for (size_t d = 0; d < dim; d++) lk_sub(&lks[from * dim + d], data[idx * dim + d]);
    assign[idx] = to;
    clusters[to]++;
    for (size_t d = 0; d < dim; d++) lk_add(&lks[to * dim + d], data[idx * dim + d]);
}

static void fill_lks(const double *data, size_t n, size_t dim, const size_t *assign, size_t k,
                     lkcount_t *lks, size_t *clusters) {
    memset(lks, 0, k * dim * sizeof(lkcount_t));
    memset(clusters, 0, k * sizeof(size_t));
    for (size_t i = 0; i < n; i++) {
        clusters[assign[i]]++;
        for (size_t d = 0; d < dim; d++) lk_add(&lks[assign[i] * dim + d], data[i * dim + d]);
    }
}

/* pseudo_mcmc.rs:704-762.  Returns NaN where the reference panics:
 *   assert!(is_valid_lk) :714-715 (a NaN in size_to_lk), LKCount::{add,sub}'s assert!(x.abs() < POS_THR) :830,:841
 *   (a value that is neither > POS_THR nor < -POS_THR nor inside the band: exactly +-POS_THR, or NaN) and
 *   assert!((max - lk).abs() < 0.0001) :759-760 (the likelihood of argmax recomputed from scratch). */
double jo_mcmc_with_filter(const double *data, size_t n, size_t dim, size_t *assign, size_t k, double cov,
                           jo_rng_t *rng) {
    double *size_to_lk = (double *)malloc((n + 1) * sizeof(double));
    int panic = 0;
    for (size_t x = 0; x <= n; x++) {
        size_to_lk[x] = jo_max_poisson_lk(x, cov, 1, k);
        if (size_to_lk[x] != size_to_lk[x]) panic = 1; /* :714-715 */
    }
    for (size_t e = 0; e < n * dim; e++) { /* every row is added once before the first proposal (:720-725) */
        const double x = data[e];
        if (!(JO_POS_THR < x) && !(x < -JO_POS_THR) && !(__builtin_fabs(x) < JO_POS_THR)) panic = 1; /* :830 */
    }
    if (panic) {
        free(size_to_lk);
        return __builtin_nan("");
    }
    size_t *clusters = (size_t *)malloc(k * sizeof(size_t));
    lkcount_t *lks = (lkcount_t *)malloc(k * dim * sizeof(lkcount_t));
    uint8_t *use = (uint8_t *)malloc(dim ? dim : 1);
    size_t *argmax = (size_t *)malloc(n * sizeof(size_t));
    fill_lks(data, n, dim, assign, k, lks, clusters);
    double lk = get_lk(lks, clusters, k, dim, size_to_lk, use);
    double max = lk;
    memcpy(argmax, assign, n * sizeof(size_t));
    size_t total = 2000 * n;
    for (size_t t = 0; t < total; t++) {
        size_t idx = (size_t)jo_gen_range_usize(rng, n);
        size_t old = assign[idx];
        size_t nw = (size_t)jo_choose_other(rng, k, old);
        flip(data, dim, assign, idx, nw, lks, clusters);
        double proposed = get_lk(lks, clusters, k, dim, size_to_lk, use);
        double diff = proposed - lk;
        if (0.0 < diff || jo_gen_bool(rng, jtk_exp(diff))) {
            lk = proposed;
            if (max < lk) {
                max = proposed;
                memcpy(argmax, assign, n * sizeof(size_t));
            }
        } else {
            flip(data, dim, assign, idx, old, lks, clusters);
        }
    }
    memcpy(assign, argmax, n * sizeof(size_t));
    /* :751-760: the likelihood of argmax from freshly filled counters must agree with the tracked maximum */
    fill_lks(data, n, dim, assign, k, lks, clusters);
    const double fresh = get_lk(lks, clusters, k, dim, size_to_lk, use);
    if (!(__builtin_fabs(max - fresh) < 0.0001)) max = __builtin_nan("");
    free(size_to_lk);
    free(clusters);
    free(lks);
    free(use);
    free(argmax);
    return max;
}

/* pseudo_mcmc.rs:381-408 */
static void get_read_lk_gains(const double *data, size_t n, size_t dim, const size_t *assign, size_t k,
                              uint8_t *use_columns, double *gain_on_read) {
    lkcount_t *lks = (lkcount_t *)malloc(k * dim * sizeof(lkcount_t));
    size_t *clusters = (size_t *)malloc(k * sizeof(size_t));
    fill_lks(data, n, dim, assign, k, lks, clusters);
    get_used_columns(lks, k, dim, use_columns);
    for (size_t i = 0; i < n; i++) {
        double s = 0.0;
        for (size_t d = 0; d < dim; d++)
            if (use_columns[d] && JO_POS_THR < lks[assign[i] * dim + d].total_gain) s += data[i * dim + d];
        gain_on_read[i] = s;
    }
    free(lks);
    free(clusters);
}

/* pseudo_mcmc.rs:353-379; out is n x k */
static void get_likelihood_gain(const double *data, size_t n, size_t dim, const size_t *assign, size_t k,
                                double *out) {
    lkcount_t *lks = (lkcount_t *)malloc(k * dim * sizeof(lkcount_t));
    size_t *clusters = (size_t *)malloc(k * sizeof(size_t));
    uint8_t *use = (uint8_t *)malloc(dim ? dim : 1);
    fill_lks(data, n, dim, assign, k, lks, clusters);
    get_used_columns(lks, k, dim, use);
    for (size_t i = 0; i < n; i++)
        for (size_t c = 0; c < k; c++) {
            double s = 0.0;
            for (size_t d = 0; d < dim; d++)
                if (use[d] && JO_POS_THR < lks[c * dim + d].total_gain) s += data[i * dim + d];
            out[i * k + c] = s;
        }
    free(lks);
    free(clusters);
    free(use);
}

/* pseudo_mcmc.rs:649-670 */
int jo_mcmc_clustering(const double *data, size_t n, size_t dim, size_t k, double cov, jo_rng_t *rng,
                       size_t *assign_out, double *score, double *lk_gains, uint8_t *used_columns) {
    size_t *cur = (size_t *)malloc(n * sizeof(size_t));
    double best = 0;
    int have = 0, rc = 0;
    for (int it = 0; it < 20; it++) {
        if (jo_kmeans(data, n, dim, k, rng, NULL, cur) != 0) {
            rc = -1;
            break;
        }
        double lk = jo_mcmc_with_filter(data, n, dim, cur, k, cov, rng);
        if (lk != lk) { /* the reference panicked inside mcmc_with_filter */
            rc = -1;
            break;
        }
        if (!have || !(lk < best)) { /* max_by: last maximum */
            best = lk;
            have = 1;
            memcpy(assign_out, cur, n * sizeof(size_t));
        }
    }
    free(cur);
    if (rc) return rc;
    get_read_lk_gains(data, n, dim, assign_out, k, used_columns, lk_gains);
    size_t *counts = (size_t *)calloc(k, sizeof(size_t));
    for (size_t i = 0; i < n; i++) counts[assign_out[i]]++;
    double cluster_lk = 0.0;
    for (size_t c = 0; c < k; c++) cluster_lk += jo_max_poisson_lk(counts[c], cov, 1, k);
    free(counts);
    *score = best - cluster_lk;
    return 0;
}

/* pseudo_mcmc.rs:673-693 */
static void use_highest_gain(const double *data, size_t n, size_t dim, size_t *assign, double *score,
                             double *lk_gains, uint8_t *used_columns) {
    double *gains = (double *)calloc(dim, sizeof(double));
    for (size_t i = 0; i < n; i++)
        for (size_t d = 0; d < dim; d++) gains[d] += jtk_fmax(data[i * dim + d], 0.0);
    size_t max_idx = 0;
    for (size_t d = 1; d < dim; d++)
        if (!(gains[d] < gains[max_idx])) max_idx = d; /* last maximum */
    for (size_t i = 0; i < n; i++) assign[i] = (0.0 < data[i * dim + max_idx]) ? 1 : 0;
    get_read_lk_gains(data, n, dim, assign, 2, used_columns, lk_gains);
    double s = 0.0;
    for (size_t i = 0; i < n; i++) s += lk_gains[i];
    *score = s;
    free(gains);
}

/* pseudo_mcmc.rs:286-306 */
static double expected_gains(const jtk_gains_t *gains, const size_t *vt_homop, const int *vt_type,
                             size_t dim, const uint8_t *prev_columns, const uint8_t *used_columns) {
    const double EXPT_GAIN_FACTOR = 0.8;
    int no_new_variants = memcmp(prev_columns, used_columns, dim) == 0;
    double expt_gain = 0.0; /* unwrap_or(0) for dim == 0 */
    for (size_t d = 0; d < dim; d++) {
        int newly_used = (!prev_columns[d]) & used_columns[d];
        int check = newly_used | no_new_variants;
        double v = check ? jo_gains_expected(gains, vt_homop[d], vt_type[d]) : 0.0000001;
        if (d == 0 || !(v < expt_gain)) expt_gain = v; /* max_by, last maximum; value only */
    }
    double r = EXPT_GAIN_FACTOR * expt_gain;
    return jtk_fmax(r, 0.1);
}

/* pseudo_mcmc.rs:213-274 */
int jo_cluster_filtered_variants(const double *variants, size_t n, size_t dim, const size_t *vt_homop,
                                 const int *vt_type, const jo_cluster_config_t *cfg, jo_rng_t *rng,
                                 size_t *assignments, double *likelihood_gains, double *score_out,
                                 size_t *k_out) {
    size_t copy_num = cfg->copy_num;
    double coverage = cfg->coverage;
    if (copy_num <= 1 || dim == 0 || n <= copy_num) {
        for (size_t i = 0; i < n; i++) {
            assignments[i] = 0;
            likelihood_gains[i] = 0.0; /* n x 1 */
        }
        *score_out = 0.0;
        *k_out = 1;
        return 0;
    }
    double per_cluster_cov = cfg->local_coverage;
    double max = 0.0;
    size_t max_k = 1;
    for (size_t i = 0; i < n; i++) assignments[i] = 0;
    uint8_t *prev_used = (uint8_t *)calloc(dim, 1);
    uint8_t *used = (uint8_t *)calloc(dim, 1), *used2 = (uint8_t *)calloc(dim, 1);
    size_t *asn = (size_t *)malloc(n * sizeof(size_t)), *asn2 = (size_t *)malloc(n * sizeof(size_t));
    double *gn = (double *)malloc(n * sizeof(double)), *gn2 = (double *)malloc(n * sizeof(double));
    size_t end = copy_num < 1 + 2 * dim ? copy_num : 1 + 2 * dim;
    size_t start = (end > 5 ? end : 5) - 3;
    int rc = 0;
    double *old_gn = (double *)calloc(n ? n : 1, sizeof(double)); /* read_lk_gains (:229): feeds the trace rows only */
    trace_row("RANGE\t%zu..=%zu", start, end);                    /* :236, {:?} of a RangeInclusive */
    for (size_t k = start; k <= end; k++) {
        double score;
        if (jo_mcmc_clustering(variants, n, dim, k, coverage, rng, asn, &score, gn, used) != 0) {
            rc = -1;
            break;
        }
        if (k == 2) {
            double hscore;
            use_highest_gain(variants, n, dim, asn2, &hscore, gn2, used2);
            if (score < hscore) {
                score = hscore;
                memcpy(asn, asn2, n * sizeof(size_t));
                memcpy(gn, gn2, n * sizeof(double));
                memcpy(used, used2, dim);
            }
        }
        double expected_gain_per_read = expected_gains(cfg->gains, vt_homop, vt_type, dim, prev_used, used);
        double expected_gain = expected_gain_per_read * per_cluster_cov + 0.1;
        if (jo_trace_sink) { /* :250-256; min_gain :276-284 (min_by: first minimum; the value only), count_improved_reads :308-312 */
            double min_gain = 1.0;
            int have_min = 0;
            for (size_t d = 0; d < dim; d++)
                if (used[d]) {
                    double v = jo_gains_expected(cfg->gains, vt_homop[d], vt_type[d]) / 3.0;
                    if (!have_min || v < min_gain) min_gain = v;
                    have_min = 1;
                }
            size_t improved = 0;
            for (size_t i = 0; i < n; i++)
                if (old_gn[i] + min_gain < gn[i]) improved++;
            char b1[64], b2[64];
            trace_row("LK\t%zu\t%s", k, fx(b1, sizeof b1, score, 3));
            trace_row("LK\t%zu\t%s\t%s\t%zu", k, fx(b1, sizeof b1, score, 3), fx(b2, sizeof b2, expected_gain, 3), improved);
        }
        if (expected_gain < score - max) {
            if (jo_trace_sink) { /* :258-262, {:?} of a Vec<usize> */
                char line[256];
                size_t w = 0;
                w += (size_t)snprintf(line + w, sizeof line - w, "COUNTS\t[");
                for (size_t c = 0; c < k; c++) {
                    size_t cnt = 0;
                    for (size_t i = 0; i < n; i++) cnt += asn[i] == c;
                    w += (size_t)snprintf(line + w, sizeof line - w, c ? ", %zu" : "%zu", cnt);
                }
                trace_row("%s]", line);
            }
            memcpy(assignments, asn, n * sizeof(size_t));
            memcpy(old_gn, gn, n * sizeof(double));
            max = score;
            max_k = k;
            memcpy(prev_used, used, dim);
        } else {
            break;
        }
    }
    if (rc == 0) {
        get_likelihood_gain(variants, n, dim, assignments, max_k, likelihood_gains);
        *score_out = max;
        *k_out = max_k;
    }
    free(old_gn);
    free(prev_used);
    free(used);
    free(used2);
    free(asn);
    free(asn2);
    free(gn);
    free(gn2);
    return rc;
}

/* pseudo_mcmc.rs:98-105 + :342-347 */
void jo_reassign_and_posterior(size_t n, size_t k, size_t *assign, double *lg) {
    for (size_t i = 0; i < n; i++) {
        double *lks = lg + i * k;
        size_t bi = 0;
        for (size_t c = 1; c < k; c++)
            if (!(lks[c] < lks[bi])) bi = c; /* max_by: last maximum */
        if (lks[assign[i]] + 0.001 < lks[bi]) assign[i] = bi;
    }
    for (size_t i = 0; i < n; i++) {
        double *xs = lg + i * k;
        double total = jo_logsumexp(xs, k);
        for (size_t c = 0; c < k; c++) xs[c] -= total;
    }
}

/* pseudo_mcmc.rs:45-68 modification_table + :141 compress + :426 filter + :180 op/homop + :70 filter_by */
size_t jo_search_variants(const uint8_t *tmpl, size_t tl, size_t n, const uint8_t *const *reads,
                          const size_t *read_len, const uint8_t *const *ops, const size_t *ops_len,
                          const uint8_t *strands, const jtk_hmm_t *fwd, const jtk_hmm_t *rev,
                          const jo_cluster_config_t *cfg, double *variants, size_t *vt_homop, int *vt_type,
                          size_t *pos_out) {
    size_t cols = NUM_ROW * (tl + 1);
    double *profiles = (double *)malloc((n ? n : 1) * cols * sizeof(double));
    for (size_t r = 0; r < n; r++) {
        const jtk_hmm_t *h = strands[r] ? fwd : rev;
        double lk = jo_phmm_modification_table(h, tmpl, tl, reads[r], read_len[r], ops[r], ops_len[r],
                                               cfg->band_width, profiles + r * cols);
        for (size_t p = 0; p < cols; p++) profiles[r * cols + p] -= lk;
    }
    jo_compress_small_gains(profiles, n, cols, tmpl, tl, cfg->gains);
    size_t cap = 3 * (cfg->copy_num > 2 ? cfg->copy_num : 2);
    double *score = (double *)malloc(cap * sizeof(double));
    size_t d = n ? jo_filter_profiles(tmpl, tl, profiles, n, strands, cfg, pos_out, score) : 0;
    size_t *homop = (size_t *)malloc((tl ? tl : 1) * sizeof(size_t));
    jo_homopolymer_length(tmpl, tl, homop);
    for (size_t j = 0; j < d; j++) { /* operation_and_homopolymer_length :180-193 */
        size_t bp;
        int dt;
        pos_to_bp_and_difftype(pos_out[j], &bp, &dt);
        vt_homop[j] = bp < tl ? homop[bp] : 0;
        vt_type[j] = dt;
    }
    for (size_t r = 0; r < n; r++)
        for (size_t j = 0; j < d; j++) variants[r * d + j] = profiles[r * cols + pos_out[j]];
    if (jo_trace_sink) /* :122-127 */
        for (size_t j = 0; j < d; j++) {
            double sum = 0.0;
            for (size_t r = 0; r < n; r++) sum += jtk_fmax(profiles[r * cols + pos_out[j]], 0.0);
            char b1[64], b2[64];
            trace_row("DUMP\t%zu\t%zu\t%zu\t%s\t%s", j, pos_out[j] / NUM_ROW, pos_out[j] % NUM_ROW, fx(b1, sizeof b1, score[j], 1),
                      fx(b2, sizeof b2, sum, 1));
        }
    free(homop);
    free(score);
    free(profiles);
    return d;
}

/* pseudo_mcmc.rs:77-107 */
int jo_clustering(const uint8_t *tmpl, size_t tl, size_t n, const uint8_t *const *reads,
                  const size_t *read_len, const uint8_t *const *ops, const size_t *ops_len,
                  const uint8_t *strands, jo_rng_t *rng, const jtk_hmm_t *fwd, const jtk_hmm_t *rev,
                  const jo_cluster_config_t *cfg, size_t *assign, double *post, double *score,
                  size_t *k_out, size_t *n_variants_out) {
    if (n_variants_out) *n_variants_out = 0;
    if (cfg->copy_num < 2) {
        for (size_t i = 0; i < n; i++) {
            assign[i] = 0;
            post[i] = 0.0; /* n x 1 */
        }
        *score = 0.0;
        *k_out = 1;
        return 0;
    }
    size_t cap = 3 * (cfg->copy_num > 2 ? cfg->copy_num : 2);
    double *variants = (double *)malloc((n ? n : 1) * cap * sizeof(double));
    size_t *vt_homop = (size_t *)malloc(cap * sizeof(size_t));
    int *vt_type = (int *)malloc(cap * sizeof(int));
    size_t *pos = (size_t *)malloc(cap * sizeof(size_t));
    size_t d = jo_search_variants(tmpl, tl, n, reads, read_len, ops, ops_len, strands, fwd, rev, cfg,
                                  variants, vt_homop, vt_type, pos);
    if (n_variants_out) *n_variants_out = d;
    int rc = jo_cluster_filtered_variants(variants, n, d, vt_homop, vt_type, cfg, rng, assign, post, score,
                                          k_out);
    if (rc == 0) jo_reassign_and_posterior(n, *k_out, assign, post);
    free(variants);
    free(vt_homop);
    free(vt_type);
    free(pos);
    return rc;
}
