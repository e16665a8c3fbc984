/* misc.c -- CPU ORACLE (test infrastructure). Restatement of the parts of haplotyper/src/misc.rs that the
 * local-clustering path uses, plus Node::recover (definitions/src/lib.rs:773-813) as used by the pile-up
 * sort (local_clustering/mod.rs:47-50).
 */
#include <stdlib.h>
#include <string.h>

#include "jtk_math.h"
#include "jtk_oracle.h"

double jo_exp(double x) { return jtk_exp(x); }
double jo_log(double x) { return jtk_log(x); }

/* misc.rs:84-92.  max_by returns the last maximum; only its value is used. */
double jo_logsumexp(const double *xs, size_t n) {
    if (n == 0) return 0.0;
    double max = xs[0];
    for (size_t i = 1; i < n; i++)
        if (!(xs[i] < max)) max = xs[i];
    double sum = 0.0;
    for (size_t i = 0; i < n; i++) sum += jtk_exp(xs[i] - max);
    return max + jtk_log(sum);
}

/* misc.rs:5-20 */
double jo_rand_index(const size_t *label, const size_t *pred, size_t n) {
    size_t both_same = 0, both_diff = 0;
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < i; j++) {
            if (label[i] == label[j] && pred[i] == pred[j])
                both_same++;
            else if (label[i] != label[j] && pred[i] != pred[j])
                both_diff++;
        }
    return (double)(both_same + both_diff) / (double)(n * (n - 1) / 2);
}

/* misc.rs:308-313 */
static double dist(const double *xs, const double *ys, size_t dim) {
    double s = 0.0;
    for (size_t d = 0; d < dim; d++) {
        double t = xs[d] - ys[d];
        s += t * t;
    }
    return s;
}

/* misc.rs:261-276; `centers` is an array of k pointers (suggest_first passes borrowed data rows).
 * min_by keeps the FIRST minimum. */
static void update_assignments(const double *data, size_t n, size_t dim, const double *const *centers,
                               size_t k, size_t *assign) {
    for (size_t i = 0; i < n; i++) {
        size_t best = 0;
        double bd = dist(data + i * dim, centers[0], dim);
        for (size_t c = 1; c < k; c++) {
            double d = dist(data + i * dim, centers[c], dim);
            if (d < bd) {
                bd = d;
                best = c;
            }
        }
        assign[i] = best;
    }
}

/* misc.rs:277-297 */
static void update_centers(const double *data, size_t n, size_t dim, double *centers, size_t *counts,
                           size_t k, const size_t *assign) {
    memset(centers, 0, k * dim * sizeof(double));
    memset(counts, 0, k * sizeof(size_t));
    for (size_t i = 0; i < n; i++) {
        double *c = centers + assign[i] * dim;
        for (size_t d = 0; d < dim; d++) c[d] += data[i * dim + d];
        counts[assign[i]]++;
    }
    for (size_t c = 0; c < k; c++)
        if (counts[c] > 0)
            for (size_t d = 0; d < dim; d++) centers[c * dim + d] /= (double)counts[c];
}

/* misc.rs:298-307 */
static double get_dist(const double *data, size_t n, size_t dim, const double *centers,
                       const size_t *assign) {
    double s = 0.0;
    for (size_t i = 0; i < n; i++) s += dist(data + i * dim, centers + assign[i] * dim, dim);
    return s;
}

/* misc.rs:315-341 */
static int suggest_first(const double *data, size_t n, size_t dim, size_t k, jo_rng_t *rng,
                         size_t *assign) {
    if (k > n) return -1; /* assert!(k <= data.len()) */
    const double **centers = (const double **)malloc(k * sizeof(*centers));
    double *dists = (double *)malloc(n * sizeof(double));
    size_t nc = 0;
    centers[nc++] = data + jo_gen_index(rng, n) * dim; /* data.choose(rng) */
    int rc = 0;
    for (size_t it = 0; it + 1 < k; it++) {
        for (size_t i = 0; i < n; i++) {
            double m = dist(data + i * dim, centers[0], dim);
            for (size_t c = 1; c < nc; c++) {
                double d = dist(data + i * dim, centers[c], dim);
                if (d < m) m = d; /* min_by: first minimum; value only */
            }
            dists[i] = m;
        }
        int64_t idx = jo_choose_weighted(rng, dists, n);
        if (idx < 0) {
            rc = -1; /* .unwrap() on WeightedError */
            break;
        }
        centers[nc++] = data + (size_t)idx * dim;
    }
    if (rc == 0) update_assignments(data, n, dim, centers, k, assign);
    free(centers);
    free(dists);
    return rc;
}

/* misc.rs:229-259 */
int jo_kmeans(const double *data, size_t n, size_t dim, size_t k, jo_rng_t *rng, double *dist_out,
              size_t *assign) {
    const double UPDATE_THR = 0.00000001;
    if (k < 1 || dim == 0) return -1;
    if (jo_gen_bool(rng, 0.5)) {
        for (size_t i = 0; i < n; i++) assign[i] = (size_t)jo_gen_range_usize(rng, k);
    } else {
        if (suggest_first(data, n, dim, k, rng, assign) != 0) return -1;
    }
    double *centers = (double *)calloc(k * dim, sizeof(double));
    size_t *counts = (size_t *)calloc(k, sizeof(size_t));
    const double **cptr = (const double **)malloc(k * sizeof(*cptr));
    for (size_t c = 0; c < k; c++) cptr[c] = centers + c * dim;
    double d = get_dist(data, n, dim, centers, assign);
    int rc = 0;
    for (;;) {
        update_centers(data, n, dim, centers, counts, k, assign);
        update_assignments(data, n, dim, cptr, k, assign);
        double nd = get_dist(data, n, dim, centers, assign);
        if (!(nd < d + UPDATE_THR)) { /* assert!(new_dist < dist + UPDATE_THR) */
            rc = -1;
            break;
        }
        if (d - nd < UPDATE_THR) break;
        d = nd;
    }
    if (dist_out) *dist_out = d;
    free(centers);
    free(counts);
    free(cptr);
    return rc;
}

/* misc.rs:188-225 kiley_op_to_ops: Match and Mismatch merge into M runs. kind: 0=M 1=D 2=I */
size_t jo_ops_to_runs(const uint8_t *ops, size_t n, uint8_t *kind, uint64_t *len) {
    if (n == 0) return 0;
    size_t nr = 0;
    uint8_t cur = 255;
    for (size_t i = 0; i < n; i++) {
        uint8_t kd = (ops[i] == JTK_OP_DEL) ? 1 : (ops[i] == JTK_OP_INS) ? 2 : 0;
        if (nr > 0 && kd == cur) {
            len[nr - 1]++;
        } else {
            kind[nr] = kd;
            len[nr] = 1;
            nr++;
            cur = kd;
        }
    }
    return nr;
}

/* mod.rs:47-50 with Node::recover (definitions/src/lib.rs:773-813): count alignment columns whose
 * symbol is not '|' -- i.e. every Ins/Del column plus every M column whose bases differ
 * (case-insensitively).  M columns compare the actual bases, not the Match/Mismatch tag: the reference's
 * cigar (definitions Ops) has no mismatch op. */
uint64_t jo_pileup_sort_key(const uint8_t *tmpl, size_t tl, const uint8_t *read, size_t rl,
                            const uint8_t *ops, size_t n_ops) {
    size_t q = 0, r = 0;
    uint64_t key = 0;
    for (size_t i = 0; i < n_ops; i++) {
        if (ops[i] == JTK_OP_DEL) {
            key++;
            r++;
        } else if (ops[i] == JTK_OP_INS) {
            key++;
            q++;
        } else {
            uint8_t a = (q < rl) ? read[q] : 0, b = (r < tl) ? tmpl[r] : 1;
            if ((a & 0xdf) != (b & 0xdf)) key++;
            q++;
            r++;
        }
    }
    return key;
}
