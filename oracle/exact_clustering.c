/* exact_clustering.c -- CPU ORACLE (test infrastructure).  Restatement of
 * haplotyper/src/local_clustering/exact_clustering.rs:7-77: the brute-force optimum the reference's own harness
 * compares the Metropolis clustering against (sandbox/src/bin/benchmark_mcmc.rs:111-122).  Used by tests/ as a
 * reference-derived bound on the chain's objective; never by the product.
 *
 * A "selection" is a bit mask over the feature columns; a candidate solution is one selection per cluster, kept
 * non-increasing (increment_one :65-77).  score(vars) = sum over reads of max over clusters of the sum of the
 * read's values on the cluster's selected columns (calc_score :53-63, get_exact_score :46-51: left-to-right sum
 * over the selected columns in column order, max_by = last maximum). */
#include <stdlib.h>

#include "jtk_oracle.h"

static double get_exact_score(size_t selection, const double *xs, size_t dim) { /* :46-51 */
    double s = 0.0;
    for (size_t i = 0; i < dim; i++)
        if ((((size_t)1 << i) & selection) != 0) s += xs[i];
    return s;
}

static double calc_score(const size_t *vars, size_t k, const double *variants, size_t n, size_t dim) { /* :53-63 */
    double total = 0.0;
    for (size_t r = 0; r < n; r++) {
        double best = 0.0;
        for (size_t c = 0; c < k; c++) {
            const double v = get_exact_score(vars[c], variants + r * dim, dim);
            if (c == 0 || !(v < best)) best = v;
        }
        total += best;
    }
    return total;
}

static void increment_one(size_t *vars, size_t max) { /* :65-77 */
    size_t idx = 0;
    while (max == vars[idx] + 1) idx++;
    vars[idx] += 1;
    for (size_t j = 0; j < idx; j++) vars[j] = vars[idx];
}

static void get_result(const size_t *vars, size_t k, const double *variants, size_t n, size_t dim,
                       size_t *assign, double *lk_gain, double *score) { /* :28-44 */
    *score = calc_score(vars, k, variants, n, dim);
    for (size_t r = 0; r < n; r++) {
        size_t max_id = 0;
        for (size_t c = 0; c < k; c++) {
            lk_gain[r * k + c] = get_exact_score(vars[c], variants + r * dim, dim);
            if (!(lk_gain[r * k + c] < lk_gain[r * k + max_id])) max_id = c; /* max_by: last maximum */
        }
        assign[r] = max_id;
    }
}

/* cluster_filtered_variants_exact (:7-26): assign[n], lk_gain[n x copy_num]; returns the score of the arg-max
 * (the third element of the reference's tuple; the fourth, k, is copy_num).  dim must be < 8*sizeof(size_t) and the
 * search is exponential: C(2^dim + copy_num - 1, copy_num) candidates. */
double jo_cluster_filtered_variants_exact(const double *variants, size_t n, size_t dim, size_t copy_num,
                                          size_t *assign, double *lk_gain) {
    size_t *sel = (size_t *)calloc(copy_num ? copy_num : 1, sizeof(size_t));
    size_t *best = (size_t *)calloc(copy_num ? copy_num : 1, sizeof(size_t));
    const size_t choises = (size_t)1 << dim;
    double max = 0.0;
    for (;;) {
        int last = 1; /* while selected_variants != last_loop: the all-(choises-1) candidate is never scored */
        for (size_t c = 0; c < copy_num; c++)
            if (sel[c] != choises - 1) last = 0;
        if (last) break;
        const double score = calc_score(sel, copy_num, variants, n, dim);
        if (max < score) {
            for (size_t c = 0; c < copy_num; c++) best[c] = sel[c];
            max = score;
        }
        increment_one(sel, choises);
    }
    double score;
    get_result(best, copy_num, variants, n, dim, assign, lk_gain, &score);
    free(sel);
    free(best);
    return score;
}
