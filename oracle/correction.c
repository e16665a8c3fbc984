/* correction.c -- CPU ORACLE (test infrastructure).  Restatement of the first consumer of the stage's posteriors,
 * `AlignmentCorrection::correct_clustering_selected` (haplotyper/src/phmm_likelihood_correction.rs:32-97) with everything
 * it calls in that file: estimate_copy_number_of_cluster :131-181, correct_chunk :184-218, adj_rand_on_biased :220-240,
 * to_context :243-261, clustering :263-328, filter_similarity :330-347, select_nth :349-354, append_posterior_probability
 * :356-367, normalize_columns :369-381, get_graph_laplacian :385-402, get_eigenvalues :405-464, alignment :466-479,
 * align_swg :482-531, sim :534-550, logit_from_lnp :553-566, supress_threshold :100-105, get_protected_clusterings
 * :108-129; misc.rs adjusted_rand_index :22-46, LogSumExp :94-140, kmeans :231-341; definitions Node::is_biased :703-709.
 * Followed line by line (iterator order, stable sorts, `max_by` = last maximum, `min_by` = first minimum).
 * Three things are NOT in the reference tree and are stood in for (parity with them is unpinned):
 *   nalgebra `symmetric_eigen` (:419)         -> include/jtk_eigen.h (cyclic Jacobi; eigenvectors up to sign)
 *   rand_xoshiro Xoroshiro128PlusPlus (:295)  -> rng.c (published algorithm)
 *   estimate_minimum_gain (:118, kiley sims)  -> the caller passes the value (`min_gain`)
 * and f64::ln_1p (:575) is jtk_log(1 + x) like every other libm call of this build.
 * A reference panic (assert!, unwrap on None, index out of bounds) fails the whole call: returns JTK_ERR_CHUNK_FAILED. */
#include <stdlib.h>
#include <string.h>

#include "jtk_eigen.h"
#include "jtk_math.h"
#include "jtk_oracle.h"

typedef struct arm_ent {
    uint64_t chunk;
    const double *post;
    size_t len;
} arm_ent_t;
typedef struct context {
    arm_ent_t *up, *down;
    size_t n_up, n_down;
    const jtk_cc_node_t *center;
    const double *center_post;
} context_t;
typedef struct view { /* the dataset as the functions below see it */
    size_t n_reads;
    const uint64_t *read_id, *node_off;
    const jtk_cc_node_t *nodes;
    const double *post;
    double **copy_numbers; /* by chunk id */
    size_t *cn_len;
    size_t max_id;
} view_t;

static int g_panic; /* thread-unsafe on purpose: the oracle runs this path single-threaded */

/* logit_from_lnp :553-566 */
static double logit_from_lnp(double lnp) {
    if (!(lnp <= 0.0)) g_panic = 1;
    const double LOWER_CUT = -80.0, UPPER_CUT = 80.0, UPPER_THR = -1.8e-35;
    if (lnp < LOWER_CUT) return LOWER_CUT;
    if (UPPER_THR < lnp) return UPPER_CUT;
    return lnp - jtk_log(1.0 + (-jtk_exp(lnp)));
}

/* sim :534-550 (LogSumExp: misc.rs:94-140) */
static double sim(const double *xs, size_t nx, const double *ys, size_t ny, const double *cps, size_t nc) {
    if (nx != nc || nx != ny) {
        g_panic = 1;
        return 0.0;
    }
    if (nc == 1) {
        double total = 0.0;
        for (size_t i = 0; i < nc; i++) total += cps[i];
        return -jtk_log(jtk_fmax(total, 1.5) - 1.0);
    }
    double accum = 0.0, mx = -__builtin_inf();
    for (size_t i = 0; i < nc; i++) {
        const double rhs = xs[i] + ys[i] - jtk_log(cps[i]);
        if (rhs < mx) {
            accum = accum + jtk_exp(rhs - mx);
        } else {
            accum = accum * jtk_exp(mx - rhs) + 1.0;
            mx = rhs;
        }
    }
    const double logp = jtk_log(accum) + mx;
    const double logit = logit_from_lnp(logp);
    if (logit == __builtin_inf() || logit == -__builtin_inf()) g_panic = 1;
    return logit;
}

static double max3(double a, double b, double c) { return jtk_fmax(jtk_fmax(a, b), c); }

/* align_swg :482-531 */
static double align_swg(const arm_ent_t *arm1, size_t len1, const arm_ent_t *arm2, size_t len2, const view_t *v) {
    const double GAP_OPEN = -0.5, GAP_EXTEND = -100.0, MISM = -100.0;
    const double lower = (double)(len1 + len2 + 2) * MISM;
    const size_t W = len2 + 1;
    double *dp = (double *)malloc((len1 + 1) * W * 3 * sizeof(double));
    for (size_t e = 0; e < (len1 + 1) * W * 3; e++) dp[e] = lower;
    for (size_t i = 1; i <= len1; i++) dp[(i * W + 0) * 3 + 2] = GAP_OPEN + (double)(i - 1) * GAP_EXTEND;
    for (size_t j = 1; j <= len2; j++) dp[(0 * W + j) * 3 + 1] = GAP_OPEN + (double)(j - 1) * GAP_EXTEND;
    dp[0] = 0.0;
    for (size_t i = 1; i <= len1; i++)
        for (size_t j = 1; j <= len2; j++) {
            const arm_ent_t *a = &arm1[i - 1], *b = &arm2[j - 1];
            double match_score = MISM;
            if (a->chunk == b->chunk) {
                if (a->chunk > v->max_id || !v->copy_numbers[a->chunk]) {
                    g_panic = 1;
                    match_score = 0.0;
                } else {
                    match_score = sim(a->post, a->len, b->post, b->len, v->copy_numbers[a->chunk], v->cn_len[a->chunk]);
                }
            }
            const double *d = &dp[((i - 1) * W + (j - 1)) * 3];
            const double mat = max3(d[0], d[1], d[2]) + match_score;
            const double *l = &dp[(i * W + (j - 1)) * 3];
            const double del2 = jtk_fmax(jtk_fmax(l[0] + GAP_OPEN, l[1] + GAP_EXTEND), l[2] + GAP_OPEN);
            const double *u = &dp[((i - 1) * W + j) * 3];
            const double del1 = jtk_fmax(jtk_fmax(u[0] + GAP_OPEN, u[1] + GAP_OPEN), u[2] + GAP_EXTEND);
            double *o = &dp[(i * W + j) * 3];
            o[0] = mat;
            o[1] = del2;
            o[2] = del1;
        }
    /* row_last.chain(column_last).max_by: the last maximum */
    double best = 0.0;
    int have = 0;
    for (size_t j = 0; j <= len2; j++) {
        const double *c = &dp[(len1 * W + j) * 3];
        const double x = max3(c[0], c[1], c[2]);
        if (!have || !(x < best)) best = x, have = 1;
    }
    for (size_t i = 0; i <= len1; i++) {
        const double *c = &dp[(i * W + len2) * 3];
        const double x = max3(c[0], c[1], c[2]);
        if (!have || !(x < best)) best = x, have = 1;
    }
    free(dp);
    return best;
}

/* alignment :466-479 */
static double alignment(const context_t *c1, const context_t *c2, const view_t *v) {
    if (c1->center->chunk != c2->center->chunk) g_panic = 1;
    const double up_aln = align_swg(c1->up, c1->n_up, c2->up, c2->n_up, v);
    const double down_aln = align_swg(c1->down, c1->n_down, c2->down, c2->n_down, v);
    const uint64_t cid = c1->center->chunk;
    const double center = sim(c1->center_post, c1->center->post_len, c2->center_post, c2->center->post_len, v->copy_numbers[cid],
                              v->cn_len[cid]);
    const double likelihood_ratio = up_aln + down_aln + center;
    return 1.0 / (1.0 + jtk_exp(-likelihood_ratio));
}

/* the full similarity matrix of one chunk (:272-285): also exported, so that the device kernel can be compared on it */
static void similarity_matrix(const context_t *ctx, size_t n, const view_t *v, double *sims) {
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) sims[i * n + j] = i == j ? 0.0 : alignment(&ctx[i], &ctx[j], v);
}

static int cmp_f64(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* filter_similarity :330-347, select_nth :349-354 */
static void filter_similarity(double *sims, size_t n, size_t len) {
    const double SMALL = 0.0000000000000001, MIN_REQ = 0.51;
    uint8_t *keep = (uint8_t *)calloc(n * n, 1);
    double *tmp = (double *)malloc((n ? n : 1) * sizeof(double));
    for (size_t i = 0; i < n; i++) {
        if (!(len < n)) { /* assert!(pivot <= len) passes for pivot == len, then sims[pivot] is out of bounds */
            g_panic = 1;
            break;
        }
        memcpy(tmp, sims + i * n, n * sizeof(double));
        qsort(tmp, n, sizeof(double), cmp_f64);
        const double threshold = jtk_fmax(tmp[len], MIN_REQ);
        for (size_t j = 0; j < n; j++)
            if (threshold <= sims[i * n + j]) keep[i * n + j] = keep[j * n + i] = 1;
    }
    for (size_t e = 0; e < n * n; e++)
        if (!keep[e]) sims[e] = SMALL;
    free(keep);
    free(tmp);
}

typedef struct eig_pair {
    double lam;
    size_t col;
} eig_pair_t;

/* misc.rs:22-46 */
static double adjusted_rand_index(const size_t *label, const size_t *pred, size_t n) {
    if (n == 0) { /* iter().max().unwrap() on an empty slice */
        g_panic = 1;
        return 0.0;
    }
    size_t lab_max = 0, pred_max = 0;
    for (size_t i = 0; i < n; i++) {
        if (label[i] > lab_max) lab_max = label[i];
        if (pred[i] > pred_max) pred_max = pred[i];
    }
    size_t *cont = (size_t *)calloc((lab_max + 1) * (pred_max + 1), sizeof(size_t));
    size_t *lab_sum = (size_t *)calloc(lab_max + 1, sizeof(size_t)), *pred_sum = (size_t *)calloc(pred_max + 1, sizeof(size_t));
    for (size_t i = 0; i < n; i++) {
        cont[label[i] * (pred_max + 1) + pred[i]]++;
        lab_sum[label[i]]++;
        pred_sum[pred[i]]++;
    }
#define CHOOSE(x) ((((x) > 1 ? (x) : 1) - 1) * (x) / 2)
    size_t lab_match = 0, pred_match = 0, both_match = 0;
    for (size_t i = 0; i <= lab_max; i++) lab_match += CHOOSE(lab_sum[i]);
    for (size_t i = 0; i <= pred_max; i++) pred_match += CHOOSE(pred_sum[i]);
    const size_t num_of_pairs = CHOOSE(n);
    for (size_t e = 0; e < (lab_max + 1) * (pred_max + 1); e++) both_match += CHOOSE(cont[e]);
#undef CHOOSE
    if (!(both_match <= (lab_match + pred_match) / 2)) g_panic = 1;
    const int64_t match_prod = (int64_t)(lab_match * pred_match);
    const int64_t denom = (int64_t)(num_of_pairs * (lab_match + pred_match) / 2) - match_prod;
    const int64_t numer = (int64_t)(num_of_pairs * both_match) - match_prod;
    free(cont);
    free(lab_sum);
    free(pred_sum);
    return (double)numer / (double)denom;
}

/* Node::is_biased, definitions/src/lib.rs:703-709 */
static int is_biased(const double *post, size_t len, double thr) {
    if (len <= 1) return 1;
    const double t = 1.0 / (double)len + thr;
    for (size_t i = 0; i < len; i++)
        if (t <= jtk_exp(post[i])) return 1;
    return 0;
}

typedef struct member { /* one (idx, read) of correct_chunk's `reads` */
    size_t read, idx;
    uint64_t cluster;
} member_t;

/* clustering :263-328 on the members of one chunk (already sorted); asn_out[n]; returns cluster_num */
static size_t clustering(const member_t *mem, size_t n, size_t k, const jtk_cc_chunk_t *chunk, const view_t *v, size_t *asn_out,
                         double *sims_out) {
    context_t *ctx = (context_t *)calloc(n ? n : 1, sizeof(context_t));
    for (size_t m = 0; m < n; m++) { /* to_context :243-261 */
        const size_t r = mem[m].read, idx = mem[m].idx;
        const jtk_cc_node_t *rn = v->nodes + v->node_off[r];
        const size_t len = (size_t)(v->node_off[r + 1] - v->node_off[r]);
        arm_ent_t *before = (arm_ent_t *)malloc((idx ? idx : 1) * sizeof(arm_ent_t));
        arm_ent_t *after = (arm_ent_t *)malloc((len - idx) * sizeof(arm_ent_t) + sizeof(arm_ent_t));
        for (size_t q = 0; q < idx; q++) { /* nodes[..idx] reversed */
            const jtk_cc_node_t *nd = &rn[idx - 1 - q];
            before[q].chunk = nd->chunk;
            before[q].post = v->post + nd->post_off;
            before[q].len = nd->post_len;
        }
        for (size_t q = idx + 1; q < len; q++) {
            after[q - idx - 1].chunk = rn[q].chunk;
            after[q - idx - 1].post = v->post + rn[q].post_off;
            after[q - idx - 1].len = rn[q].post_len;
        }
        const int fwd = rn[idx].is_forward != 0;
        ctx[m].up = fwd ? before : after;
        ctx[m].n_up = fwd ? idx : len - idx - 1;
        ctx[m].down = fwd ? after : before;
        ctx[m].n_down = fwd ? len - idx - 1 : idx;
        ctx[m].center = &rn[idx];
        ctx[m].center_post = v->post + rn[idx].post_off;
    }
    double *sims = (double *)malloc((n ? n * n : 1) * sizeof(double));
    similarity_matrix(ctx, n, v, sims);
    if (sims_out) memcpy(sims_out, sims, n * n * sizeof(double));
    size_t cluster_num = 0;
    if (chunk->copy_num == 0) g_panic = 1;
    if (!g_panic) {
        const size_t cov_per_copy = n - n / chunk->copy_num / 4;
        filter_similarity(sims, n, cov_per_copy);
    }
    if (!g_panic) {
        /* get_graph_laplacian :385-402 */
        double *rowsum = (double *)malloc(n * sizeof(double)), *sq_inv = (double *)malloc(n * sizeof(double));
        double *lap = (double *)malloc(n * n * sizeof(double)), *vec = (double *)malloc(n * n * sizeof(double));
        for (size_t i = 0; i < n; i++) {
            double s = 0.0;
            for (size_t j = 0; j < n; j++) s += sims[i * n + j];
            rowsum[i] = s;
            sq_inv[i] = __builtin_sqrt(1.0 / s);
        }
        for (size_t i = 0; i < n; i++)
            for (size_t j = 0; j < n; j++) lap[i * n + j] = j == i ? 1.0 : -sims[i * n + j] * sq_inv[i] * sq_inv[j];
        /* get_eigenvalues :405-464 */
        if (n == 0) g_panic = 1;
        jtk_symmetric_eigen(lap, n, vec);
        eig_pair_t *ep = (eig_pair_t *)malloc((n ? n : 1) * sizeof(eig_pair_t));
        for (size_t i = 0; i < n; i++) {
            ep[i].lam = lap[i * n + i];
            ep[i].col = i;
        }
        for (size_t i = 1; i < n; i++) { /* stable sort by |lambda| ascending */
            eig_pair_t x = ep[i];
            size_t j = i;
            while (j > 0 && __builtin_fabs(ep[j - 1].lam) > __builtin_fabs(x.lam)) {
                ep[j] = ep[j - 1];
                j--;
            }
            ep[j] = x;
        }
        size_t pick_k = 0;
        while (pick_k < n && ep[pick_k].lam < 0.2) pick_k++; /* EIGEN_THR, take_while */
        if (pick_k == 0) g_panic = 1;
        if (!g_panic) {
            const size_t pl = chunk->cluster_num; /* every node of the chunk carries cluster_num posteriors */
            const size_t dim = pick_k + pl;
            double *feat = (double *)malloc(n * dim * sizeof(double));
            for (size_t i = 0; i < n; i++) {
                const double d = __builtin_sqrt(1.0 / rowsum[i]);
                for (size_t j = 0; j < pick_k; j++) feat[i * dim + j] = vec[i * n + ep[j].col] * d;
                /* append_posterior_probability :356-367 */
                const jtk_cc_node_t *nd = ctx[i].center;
                if (nd->post_len != pl) {
                    g_panic = 1; /* rows of different length: normalize_columns indexes out of bounds or kmeans asserts */
                    break;
                }
                double tmp[64];
                for (size_t c = 0; c < pl && c < 64; c++) tmp[c] = ctx[i].center_post[c];
                const double total = jo_logsumexp(tmp, pl);
                for (size_t c = 0; c < pl; c++) feat[i * dim + pick_k + c] = jtk_exp(ctx[i].center_post[c] - total);
            }
            if (!g_panic) {
                /* normalize_columns :369-381 */
                for (size_t c = 0; c < dim; c++) {
                    double s = 0.0;
                    for (size_t i = 0; i < n; i++) s += feat[i * dim + c] * feat[i * dim + c];
                    s = __builtin_sqrt(s);
                    for (size_t i = 0; i < n; i++) feat[i * dim + c] /= s;
                }
                jo_rng_t rng;
                jo_rng128pp_seed_from_u64(&rng, chunk->id * (uint64_t)k); /* :299-301 */
                cluster_num = k < pick_k ? k : pick_k;
                size_t *cur = (size_t *)malloc(n * sizeof(size_t));
                double best = 0.0;
                int have = 0;
                for (int it = 0; it < 20; it++) { /* :303-307: min_by keeps the first minimum */
                    double dist = 0.0;
                    if (jo_kmeans(feat, n, dim, cluster_num, &rng, &dist, cur) != 0) {
                        g_panic = 1;
                        break;
                    }
                    if (!have || dist < best) {
                        best = dist;
                        have = 1;
                        memcpy(asn_out, cur, n * sizeof(size_t));
                    }
                }
                free(cur);
            }
            free(feat);
        }
        free(ep);
        free(rowsum);
        free(sq_inv);
        free(lap);
        free(vec);
    }
    for (size_t m = 0; m < n; m++) {
        const int fwd = ctx[m].center->is_forward != 0;
        free(fwd ? ctx[m].up : ctx[m].down);
        free(fwd ? ctx[m].down : ctx[m].up);
    }
    free(ctx);
    free(sims);
    return cluster_num;
}

static double round_half_away(double x) { return x < 0.0 ? -__builtin_floor(-x + 0.5) : __builtin_floor(x + 0.5); }

int jo_correct_clustering(size_t n_reads, const uint64_t *read_id, const uint64_t *node_off, const jtk_cc_node_t *nodes,
                          const double *posteriors, size_t n_chunks, jtk_cc_chunk_t *chunks, size_t n_selected,
                          const uint64_t *selection, double haploid_coverage, double min_gain, uint64_t *cluster_out,
                          uint8_t *touched, double *ari_out, double *sims_first) {
    g_panic = 0;
    const size_t n_nodes = (size_t)node_off[n_reads];
    for (size_t e = 0; e < n_nodes; e++) {
        cluster_out[e] = nodes[e].cluster;
        touched[e] = 0;
    }
    if (n_chunks == 0) return JTK_ERR_CHUNK_FAILED; /* .max().unwrap() :131 */
    view_t v;
    memset(&v, 0, sizeof v);
    v.n_reads = n_reads;
    v.read_id = read_id;
    v.node_off = node_off;
    v.nodes = nodes;
    v.post = posteriors;
    for (size_t c = 0; c < n_chunks; c++)
        if (chunks[c].id > v.max_id) v.max_id = chunks[c].id;
    /* ---- estimate_copy_number_of_cluster :131-181 */
    size_t *cps = (size_t *)calloc(v.max_id + 1, sizeof(size_t)), *cls = (size_t *)calloc(v.max_id + 1, sizeof(size_t));
    v.copy_numbers = (double **)calloc(v.max_id + 1, sizeof(double *));
    v.cn_len = (size_t *)calloc(v.max_id + 1, sizeof(size_t));
    for (size_t c = 0; c < n_chunks; c++) {
        cps[chunks[c].id] = chunks[c].copy_num;
        cls[chunks[c].id] = chunks[c].cluster_num;
    }
    for (size_t id = 0; id <= v.max_id; id++) {
        v.copy_numbers[id] = (double *)calloc(cls[id] ? cls[id] : 1, sizeof(double));
        v.cn_len[id] = cls[id];
    }
    for (size_t e = 0; e < n_nodes && !g_panic; e++) {
        const jtk_cc_node_t *nd = &nodes[e];
        if (nd->chunk > v.max_id) {
            g_panic = 1; /* obs_counts[chunk] out of bounds */
            break;
        }
        const double *p = posteriors + nd->post_off;
        double tmp[64];
        for (size_t c = 0; c < nd->post_len && c < 64; c++) tmp[c] = p[c];
        const double total = jo_logsumexp(tmp, nd->post_len);
        const size_t m = nd->post_len < cls[nd->chunk] ? nd->post_len : cls[nd->chunk]; /* zip */
        for (size_t c = 0; c < m; c++) v.copy_numbers[nd->chunk][c] += jtk_exp(p[c] - total);
    }
    for (size_t id = 0; id <= v.max_id && !g_panic; id++) {
        double *obs = v.copy_numbers[id];
        const size_t kk = cls[id], total_cp = cps[id];
        double *est = (double *)malloc((kk ? kk : 1) * sizeof(double));
        double sumf = 0.0;
        for (size_t c = 0; c < kk; c++) {
            est[c] = jtk_fmax(round_half_away(obs[c] / haploid_coverage), 1.0);
            sumf += est[c];
        }
        const size_t sum = (size_t)round_half_away(sumf);
        for (size_t it = sum < total_cp ? sum : total_cp; it < total_cp; it++) {
            size_t arg = 0;
            double best = 0.0;
            int have = 0;
            for (size_t c = 0; c < kk; c++) {
                const double now = (obs[c] - est[c] * haploid_coverage) * (obs[c] - est[c] * haploid_coverage);
                const double next = (obs[c] - (est[c] + 1.0) * haploid_coverage) * (obs[c] - (est[c] + 1.0) * haploid_coverage);
                const double d = now - next;
                if (!have || !(d < best)) best = d, arg = c, have = 1;
            }
            if (have) est[arg] += 1.0;
        }
        memcpy(obs, est, kk * sizeof(double));
        free(est);
    }
    /* ---- correct_chunk :184-218 for every selected chunk with more than one cluster, in selected_chunks order */
    typedef struct result {
        member_t *mem;
        size_t *asn;
        size_t n, k, chunk;
        double ari;
    } result_t;
    result_t *res = (result_t *)calloc(n_chunks, sizeof(result_t));
    size_t n_res = 0;
    for (size_t c = 0; c < n_chunks && !g_panic; c++) {
        int sel = 0;
        for (size_t q = 0; q < n_selected; q++) sel |= selection[q] == chunks[c].id;
        if (!(1 < chunks[c].cluster_num && sel)) continue;
        size_t n = 0;
        for (size_t e = 0; e < n_nodes; e++) n += nodes[e].chunk == chunks[c].id;
        member_t *mem = (member_t *)malloc((n ? n : 1) * sizeof(member_t));
        size_t w = 0;
        for (size_t r = 0; r < n_reads; r++)
            for (size_t idx = 0; idx < (size_t)(node_off[r + 1] - node_off[r]); idx++)
                if (nodes[node_off[r] + idx].chunk == chunks[c].id) {
                    mem[w].read = r;
                    mem[w].idx = idx;
                    mem[w].cluster = nodes[node_off[r] + idx].cluster;
                    w++;
                }
        for (size_t i = 1; i < n; i++) { /* sort_by_cached_key: stable, by the node's cluster */
            member_t x = mem[i];
            size_t j = i;
            while (j > 0 && mem[j - 1].cluster > x.cluster) {
                mem[j] = mem[j - 1];
                j--;
            }
            mem[j] = x;
        }
        size_t *asn = (size_t *)calloc(n ? n : 1, sizeof(size_t));
        const size_t k = clustering(mem, n, chunks[c].cluster_num, &chunks[c], &v, asn, n_res == 0 ? sims_first : NULL);
        double ari = 0.0;
        if (!g_panic) { /* adj_rand_on_biased :220-240 */
            size_t *prev = (size_t *)malloc((n ? n : 1) * sizeof(size_t)), *pb = (size_t *)malloc((n ? n : 1) * sizeof(size_t)),
                   *ab = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
            size_t nb = 0;
            for (size_t i = 0; i < n; i++) {
                const jtk_cc_node_t *nd = &nodes[node_off[mem[i].read] + mem[i].idx];
                prev[i] = (size_t)nd->cluster;
                if (is_biased(posteriors + nd->post_off, nd->post_len, 0.2)) {
                    pb[nb] = (size_t)nd->cluster;
                    ab[nb] = asn[i];
                    nb++;
                }
            }
            (void)adjusted_rand_index(prev, asn, n); /* adj_raw: evaluated (it can panic), only logged */
            const double adj = adjusted_rand_index(pb, ab, nb);
            ari = adj != adj ? 1.0 : adj;
            free(prev);
            free(pb);
            free(ab);
        }
        res[n_res].mem = mem;
        res[n_res].asn = asn;
        res[n_res].n = n;
        res[n_res].k = k;
        res[n_res].chunk = c;
        res[n_res].ari = ari;
        if (ari_out) ari_out[c] = ari;
        n_res++;
    }
    if (!g_panic) {
        /* get_protected_clusterings :108-129 */
        uint8_t *prot = (uint8_t *)calloc(n_chunks, 1);
        for (size_t c = 0; c < n_chunks; c++) {
            size_t cov = 0;
            for (size_t e = 0; e < n_nodes; e++) cov += nodes[e].chunk == chunks[c].id;
            if (cov == 0) continue; /* coverage.get(&c.id)? */
            const double cl = (double)chunks[c].cluster_num;
            const double improve_frac = (cl - 1.0) / cl;
            prot[c] = (double)cov * improve_frac * min_gain < chunks[c].score;
        }
        /* supress_threshold :100-105 */
        double *aris = (double *)malloc((n_res ? n_res : 1) * sizeof(double));
        for (size_t i = 0; i < n_res; i++) aris[i] = res[i].ari;
        qsort(aris, n_res, sizeof(double), cmp_f64);
        const size_t pick = (size_t)__builtin_ceil((double)n_res * 0.05);
        const double supress_cluster = pick < n_res ? aris[pick] : 1.0;
        free(aris);
        /* :46-96 */
        for (size_t i = 0; i < n_res && !g_panic; i++) {
            jtk_cc_chunk_t *chunk = &chunks[res[i].chunk];
            const size_t cluster_num = res[i].k;
            if (!(cluster_num <= chunk->copy_num)) {
                g_panic = 1;
                break;
            }
            const int supress = cluster_num == 1 || res[i].ari < supress_cluster;
            if (supress && prot[res[i].chunk]) continue;
            chunk->cluster_num = supress ? 1 : (uint32_t)cluster_num;
            for (size_t m = 0; m < res[i].n; m++) {
                const size_t e = (size_t)node_off[res[i].mem[m].read] + res[i].mem[m].idx;
                cluster_out[e] = supress ? 0 : (uint64_t)res[i].asn[m];
                touched[e] = 1;
            }
        }
        free(prot);
    }
    for (size_t i = 0; i < n_res; i++) {
        free(res[i].mem);
        free(res[i].asn);
    }
    free(res);
    for (size_t id = 0; id <= v.max_id; id++) free(v.copy_numbers[id]);
    free(v.copy_numbers);
    free(v.cn_len);
    free(cps);
    free(cls);
    return g_panic ? JTK_ERR_CHUNK_FAILED : 0;
}

/* test hooks: the two stand-ins of this file that have an independent check (tests/test_independent_checks.py compares them
 * with numpy.linalg.eigh and sklearn's adjusted_rand_score) */
void jo_symmetric_eigen(double *a, size_t n, double *v) { jtk_symmetric_eigen(a, n, v); }
double jo_adjusted_rand_index(const size_t *label, const size_t *pred, size_t n) { return adjusted_rand_index(label, pred, n); }
