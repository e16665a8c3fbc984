/* normalize.c -- CPU ORACLE (test infrastructure). haplotyper/src/local_clustering/normalize.rs */
#include <stdlib.h>
#include <string.h>

#include "jtk_oracle.h"

/* normalize.rs:54-63 */
void jo_reorder_f64(double *xs, uint64_t *indices, size_t n) {
    for (size_t i = 0; i < n; i++) {
        while (indices[i] != i) {
            size_t to = (size_t)indices[i];
            double t = xs[i];
            xs[i] = xs[to];
            xs[to] = t;
            uint64_t u = indices[i];
            indices[i] = indices[to];
            indices[to] = u;
        }
    }
}
void jo_reorder_i64(int64_t *xs, uint64_t *indices, size_t n) {
    for (size_t i = 0; i < n; i++) {
        while (indices[i] != i) {
            size_t to = (size_t)indices[i];
            int64_t t = xs[i];
            xs[i] = xs[to];
            xs[to] = t;
            uint64_t u = indices[i];
            indices[i] = indices[to];
            indices[to] = u;
        }
    }
}

/* normalize.rs:26-49 for one pile-up: counts per cluster, stable sort ascending by count then reverse
 * (=> descending count; among equal counts the LARGER old index comes first), relabel, permute. */
void jo_normalize_pileup(size_t n, size_t cluster_num, uint64_t *cluster, double *post, size_t stride) {
    if (cluster_num == 0) return;
    uint64_t *from = (uint64_t *)malloc(cluster_num * sizeof(uint64_t));
    uint32_t *cnt = (uint32_t *)calloc(cluster_num, sizeof(uint32_t));
    for (size_t c = 0; c < cluster_num; c++) from[c] = c;
    for (size_t i = 0; i < n; i++) cnt[cluster[i]]++;
    /* stable insertion sort by count ascending */
    for (size_t a = 1; a < cluster_num; a++) {
        uint64_t f = from[a];
        size_t b = a;
        while (b > 0 && cnt[from[b - 1]] > cnt[f]) {
            from[b] = from[b - 1];
            b--;
        }
        from[b] = f;
    }
    for (size_t a = 0; a < cluster_num / 2; a++) { /* reverse */
        uint64_t t = from[a];
        from[a] = from[cluster_num - 1 - a];
        from[cluster_num - 1 - a] = t;
    }
    uint64_t *mapsto = (uint64_t *)malloc(cluster_num * sizeof(uint64_t));
    uint64_t *indices = (uint64_t *)malloc(cluster_num * sizeof(uint64_t));
    for (size_t to = 0; to < cluster_num; to++) mapsto[from[to]] = to;
    for (size_t i = 0; i < n; i++) {
        memcpy(indices, mapsto, cluster_num * sizeof(uint64_t));
        cluster[i] = mapsto[cluster[i]];
        jo_reorder_f64(post + i * stride, indices, cluster_num);
    }
    free(from);
    free(cnt);
    free(mapsto);
    free(indices);
}
