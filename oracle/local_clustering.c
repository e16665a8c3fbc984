/* local_clustering.c -- CPU ORACLE (test infrastructure). haplotyper/src/local_clustering/mod.rs:86-123
 * (clustering_on_pileup) and flat-array batch drivers with the C-ABI's layout.  The OpenMP loop over
 * chunks mirrors `pileups.into_par_iter()` (mod.rs:64-72); results do not depend on the schedule because
 * every chunk seeds its own RNG from its id (mod.rs:97).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "jtk_math.h"
#include "jtk_oracle.h"

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

/* ReadType::band_width (definitions/src/lib.rs:201-210) */
static size_t band_width_of(double frac, size_t len) { return (size_t)ceil((double)len * frac); }


/* ---- clustering_recursive (mod.rs:125-189) ------------------------------------------------------------ */
#define UPPER_COPY_NUM 8 /* mod.rs:85 */
#define BRANCH_NUM 4     /* mod.rs:139 */

/* estim_copy_num (mod.rs:223-243): every cluster gets one copy, each remaining copy goes to the cluster
 * whose read count is farthest from coverage * copies (max_by keeps the LAST maximum). */
static void estim_copy_num(const size_t *asn, size_t n, size_t k, size_t copy_num, double coverage,
                           size_t *copy_numbers) {
    double counts[BRANCH_NUM] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) counts[asn[i]] += 1.0;
    for (size_t c = 0; c < k; c++) copy_numbers[c] = 1;
    for (size_t it = k; it < copy_num; it++) {
        size_t arg = 0;
        double best = 0.0;
        for (size_t c = 0; c < k; c++) {
            double d = counts[c] - coverage * (double)copy_numbers[c];
            double v = d * d; /* powi(2) */
            if (c == 0 || !(v < best)) {
                best = v;
                arg = c;
            }
        }
        copy_numbers[arg] += 1;
    }
}

/* One call of clustering_recursive.  assign[n]; *post_out = malloc'd n x (*k_out) log-posteriors.
 * ops are read-only here (the split branch polishes clones, mod.rs:152-155). Returns 0 or a jtk status. */
static int clustering_recursive(const jtk_lc_params_t *params, const uint8_t *cons, size_t cl, size_t n,
                                const uint8_t *const *reads, const size_t *read_len, uint8_t *const *ops,
                                const size_t *ops_len, size_t ops_cap, const uint8_t *strands, jo_rng_t *rng,
                                const jo_cluster_config_t *cfg, size_t *assign, double **post_out,
                                double *score_out, size_t *k_out, size_t *nv_out) {
    *post_out = NULL;
    if (cfg->copy_num < UPPER_COPY_NUM) {
        size_t kcap = cfg->copy_num > 1 ? cfg->copy_num : 1;
        double *pk = (double *)malloc((n ? n : 1) * kcap * sizeof(double));
        size_t nv = 0;
        int rc = jo_clustering(cons, cl, n, reads, read_len, (const uint8_t *const *)ops, ops_len, strands, rng,
                               &params->forward, &params->reverse, cfg, assign, pk, score_out, k_out, &nv);
        if (nv_out) *nv_out = nv;
        if (rc != 0) {
            free(pk);
            return JTK_ERR_CHUNK_FAILED;
        }
        *post_out = pk;
        return 0;
    }
    jo_cluster_config_t rec = *cfg;
    rec.copy_num = BRANCH_NUM; /* mod.rs:140-141 */
    double *pk = (double *)malloc((n ? n : 1) * BRANCH_NUM * sizeof(double));
    double score = 0;
    size_t k = 1, nv = 0;
    int rc = jo_clustering(cons, cl, n, reads, read_len, (const uint8_t *const *)ops, ops_len, strands, rng,
                           &params->forward, &params->reverse, &rec, assign, pk, &score, &k, &nv);
    if (nv_out) *nv_out = nv;
    if (rc != 0) {
        free(pk);
        return JTK_ERR_CHUNK_FAILED;
    }
    if (k <= 1) { /* mod.rs:146-148 */
        *post_out = pk;
        *score_out = score;
        *k_out = k;
        return 0;
    }
    size_t copy_numbers[BRANCH_NUM];
    estim_copy_num(assign, n, k, cfg->copy_num, cfg->coverage, copy_numbers);
    size_t *sub_asn[BRANCH_NUM] = {0};
    double *sub_post[BRANCH_NUM] = {0};
    size_t sub_k[BRANCH_NUM] = {0};
    double sub_scores = 0.0;
    int status = 0;
    for (size_t c = 0; c < k && status == 0; c++) {
        /* filter_sub_clusters (mod.rs:198-221) */
        size_t m = 0;
        for (size_t i = 0; i < n; i++) m += assign[i] == c;
        const uint8_t **sreads = (const uint8_t **)malloc((m ? m : 1) * sizeof(*sreads));
        size_t *srlen = (size_t *)malloc((m ? m : 1) * sizeof(size_t));
        uint8_t **sops = (uint8_t **)malloc((m ? m : 1) * sizeof(*sops));
        size_t *solen = (size_t *)malloc((m ? m : 1) * sizeof(size_t));
        uint8_t *sstr = (uint8_t *)malloc(m ? m : 1);
        size_t j = 0;
        for (size_t i = 0; i < n; i++) {
            if (assign[i] != c) continue;
            sreads[j] = reads[i];
            srlen[j] = read_len[i];
            solen[j] = ops_len[i];
            sops[j] = (uint8_t *)malloc(ops_cap + 8);
            memcpy(sops[j], ops[i], ops_len[i]);
            sstr[j] = strands[i];
            j++;
        }
        sub_asn[c] = (size_t *)malloc((m ? m : 1) * sizeof(size_t));
        double sc = 0.0;
        if (copy_numbers[c] < 2) {
            /* clustering() returns before it looks at the consensus (pseudo_mcmc.rs:86-88) and draws nothing:
             * the polish of mod.rs:153-155 cannot reach the result, so it is not run here */
            for (size_t i = 0; i < m; i++) sub_asn[c][i] = 0;
            sub_post[c] = (double *)calloc(m ? m : 1, sizeof(double));
            sub_k[c] = 1;
        } else {
            /* HMMPolishConfig::new(band_width, seqs.len(), 0) with the clustering config's band (mod.rs:153) */
            size_t ccap = cl + cl / 4 + 64;
            uint8_t *scons = (uint8_t *)malloc(ccap + 8);
            int64_t scl = jo_phmm_polish(&params->forward, &params->reverse, cons, cl, m, sreads, srlen, sops, solen,
                                         ops_cap, sstr, cfg->band_width, m, 0, scons, ccap, NULL);
            if (scl < 0)
                status = JTK_ERR_CHUNK_FAILED;
            else {
                jo_cluster_config_t sub = *cfg;
                sub.copy_num = copy_numbers[c]; /* mod.rs:156-157 */
                status = clustering_recursive(params, scons, (size_t)scl, m, sreads, srlen, sops, solen, ops_cap,
                                              sstr, rng, &sub, sub_asn[c], &sub_post[c], &sc, &sub_k[c], NULL);
            }
            free(scons);
        }
        sub_scores += sc;
        for (size_t i = 0; i < m; i++) free(sops[i]);
        free(sreads);
        free(srlen);
        free(sops);
        free(solen);
        free(sstr);
    }
    if (status == 0) {
        /* merge (mod.rs:161-187) */
        size_t offsets[BRANCH_NUM], total = 0, pointers[BRANCH_NUM] = {0, 0, 0, 0};
        for (size_t c = 0; c < k; c++) {
            offsets[c] = total;
            total += sub_k[c];
        }
        double *merged = (double *)malloc((n ? n : 1) * total * sizeof(double));
        for (size_t i = 0; i < n && status == 0; i++) {
            size_t a = assign[i], pt = pointers[a]++;
            const double *in_ps = sub_post[a] + pt * sub_k[a];
            double *po = merged + i * total;
            size_t w = 0;
            for (size_t c = 0; c < k; c++) {
                double lk = pk[i * k + c] - jtk_log((double)sub_k[c]);
                for (size_t t = 0; t < sub_k[c]; t++) po[w++] = lk;
            }
            for (size_t t = 0; t < sub_k[a]; t++) po[t + offsets[a]] += in_ps[t] + jtk_log((double)sub_k[a]);
            double sum = 0.0;
            for (size_t t = 0; t < total; t++) sum += jtk_exp(po[t]);
            if (!(fabs(1.0 - sum) < 0.0001)) status = JTK_ERR_CHUNK_FAILED; /* assert, mod.rs:184 */
            assign[i] = offsets[a] + sub_asn[a][pt];
        }
        if (status == 0) {
            *post_out = merged;
            *score_out = sub_scores + score;
            *k_out = total;
        } else
            free(merged);
    }
    for (size_t c = 0; c < BRANCH_NUM; c++) {
        free(sub_asn[c]);
        free(sub_post[c]);
    }
    free(pk);
    return status;
}

int jo_clustering_on_pileup(const jtk_lc_params_t *params, uint64_t chunk_id, size_t copy_num,
                            const uint8_t *tmpl, size_t tl, size_t n, const uint8_t *const *reads,
                            const size_t *read_len, uint8_t **ops, size_t *ops_len, size_t ops_cap,
                            const uint8_t *strands, int skip_polish, size_t *assign, double *post,
                            size_t post_stride, uint8_t *cons, size_t cons_cap, jo_chunk_result_t *res) {
    memset(res, 0, sizeof *res);
    double t0 = now_ms();
    size_t band_width = band_width_of(params->band_frac, tl);
    jo_rng_t rng;
    jo_rng_seed_from_u64(&rng, chunk_id * 3490ULL); /* mod.rs:97 */
    int64_t cl;
    if (skip_polish) {
        if (tl > cons_cap) return JTK_ERR_INVALID_ARG;
        memcpy(cons, tmpl, tl);
        cl = (int64_t)tl;
    } else {
        /* HMMPolishConfig::new(band_width / 2, seqs.len(), 3)  (mod.rs:105) */
        cl = jo_phmm_polish(&params->forward, &params->reverse, tmpl, tl, n, reads, read_len, ops, ops_len,
                            ops_cap, strands, band_width / 2, n, 3, cons, cons_cap, &res->polish_rounds);
        if (cl < 0) return JTK_ERR_CHUNK_FAILED;
    }
    double t1 = now_ms();
    /* mod.rs:108-111 */
    double per_cluster_cov;
    if (copy_num <= 2)
        per_cluster_cov = (double)n / (double)copy_num;
    else {
        per_cluster_cov = (double)n / (double)copy_num;
        if (!(per_cluster_cov > params->haploid_coverage)) per_cluster_cov = params->haploid_coverage;
    }
    jo_cluster_config_t cfg;
    cfg.band_width = band_width / 2; /* mod.rs:112 */
    cfg.gains = &params->gains;
    cfg.coverage = params->haploid_coverage;
    cfg.copy_num = copy_num;
    cfg.local_coverage = per_cluster_cov;
    double *pk = NULL;
    double score = 0;
    size_t k = 1, nv = 0;
    int rc = clustering_recursive(params, cons, (size_t)cl, n, reads, read_len, ops, ops_len, ops_cap, strands, &rng,
                                  &cfg, assign, &pk, &score, &k, &nv); /* mod.rs:113-114 */
    if (rc == 0) {
        for (size_t i = 0; i < n; i++) {
            for (size_t c = 0; c < post_stride; c++) post[i * post_stride + c] = 0.0;
            for (size_t c = 0; c < k && c < post_stride; c++) post[i * post_stride + c] = pk[i * k + c];
        }
    }
    free(pk);
    res->score = score;
    res->cluster_num = k;
    res->cons_len = (size_t)cl;
    res->n_variants = nv;
    res->polish_ms = t1 - t0;
    res->elapsed_ms = now_ms() - t0;
    return rc == 0 ? 0 : JTK_ERR_CHUNK_FAILED;
}

int jo_cluster_chunks(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                      const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                      const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, int skip_polish,
                      uint32_t *label, double *log_post, uint32_t post_stride, jtk_lc_result_t *result,
                      uint8_t *cons_out, uint64_t *cons_off, uint64_t cons_cap, uint8_t *ops_out,
                      uint64_t *ops_out_off, uint64_t ops_cap, int n_threads, double *record_ms) {
    (void)cons_cap;
    (void)ops_cap;
    int any_fail = 0;
    /* per-chunk scratch results, then a serial pass lays consensus / ops out contiguously */
    uint8_t **cons_tmp = (uint8_t **)calloc(n_chunks, sizeof(uint8_t *));
    uint8_t ***ops_tmp = (uint8_t ***)calloc(n_chunks, sizeof(uint8_t **));
    size_t **ops_len_tmp = (size_t **)calloc(n_chunks, sizeof(size_t *));
    size_t *cons_len = (size_t *)calloc(n_chunks, sizeof(size_t));
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (long ci = 0; ci < (long)n_chunks; ci++) {
        const jtk_lc_chunk_t *ch = &chunks[ci];
        size_t n = ch->n_reads, tl = (size_t)ch->tmpl_len;
        const uint8_t **reads = (const uint8_t **)malloc((n ? n : 1) * sizeof(*reads));
        size_t *rlen = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
        uint8_t **rops = (uint8_t **)malloc((n ? n : 1) * sizeof(*rops));
        size_t *olen = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
        size_t *assign = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
        size_t slack = tl / 4 + 64;
        size_t max_ops = 0;
        for (size_t r = 0; r < n; r++) {
            size_t g = (size_t)ch->read_first + r;
            size_t l = (size_t)(ops_off[g + 1] - ops_off[g]);
            if (l > max_ops) max_ops = l;
        }
        size_t ocap = max_ops + slack;
        for (size_t r = 0; r < n; r++) {
            size_t g = (size_t)ch->read_first + r;
            reads[r] = read_bases + read_off[g];
            rlen[r] = (size_t)(read_off[g + 1] - read_off[g]);
            olen[r] = (size_t)(ops_off[g + 1] - ops_off[g]);
            rops[r] = (uint8_t *)malloc(ocap + 8);
            memcpy(rops[r], ops + ops_off[g], olen[r]);
        }
        size_t ccap = tl + slack;
        uint8_t *cons = (uint8_t *)malloc(ccap + 8);
        jo_chunk_result_t cr;
        int rc = jo_clustering_on_pileup(params, ch->chunk_id, ch->copy_num, tmpl_bases + ch->tmpl_off, tl, n,
                                         reads, rlen, rops, olen, ocap, strand + ch->read_first, skip_polish,
                                         assign, log_post + (size_t)ch->read_first * post_stride,
                                         post_stride, cons, ccap, &cr);
        result[ci].score = cr.score;
        result[ci].cluster_num = (uint32_t)cr.cluster_num;
        result[ci].status = rc;
        result[ci].polish_rounds = cr.polish_rounds;
        result[ci].n_variants = (uint32_t)cr.n_variants;
        if (record_ms) {
            record_ms[2 * ci] = cr.elapsed_ms;
            record_ms[2 * ci + 1] = cr.polish_ms;
        }
        if (rc != 0) {
#pragma omp atomic write
            any_fail = 1;
            for (size_t r = 0; r < n; r++) label[ch->read_first + r] = 0;
        } else {
            for (size_t r = 0; r < n; r++) label[ch->read_first + r] = (uint32_t)assign[r];
        }
        cons_tmp[ci] = cons;
        cons_len[ci] = rc == 0 ? cr.cons_len : 0;
        ops_tmp[ci] = rops;
        ops_len_tmp[ci] = olen;
        free(reads);
        free(rlen);
        free(assign);
    }
    uint64_t co = 0, oo = 0;
    for (size_t ci = 0; ci < n_chunks; ci++) {
        const jtk_lc_chunk_t *ch = &chunks[ci];
        if (cons_off) cons_off[ci] = co;
        if (cons_out && cons_off) memcpy(cons_out + co, cons_tmp[ci], cons_len[ci]);
        co += cons_len[ci];
        for (size_t r = 0; r < ch->n_reads; r++) {
            size_t g = (size_t)ch->read_first + r;
            size_t l = result[ci].status == 0 ? ops_len_tmp[ci][r] : 0;
            if (ops_out_off) ops_out_off[g] = oo;
            if (ops_out && ops_out_off) memcpy(ops_out + oo, ops_tmp[ci][r], l);
            oo += l;
            if (ops_out_off) ops_out_off[g + 1] = oo;
            free(ops_tmp[ci][r]);
        }
        free(ops_tmp[ci]);
        free(ops_len_tmp[ci]);
        free(cons_tmp[ci]);
    }
    if (cons_off) cons_off[n_chunks] = co;
    free(cons_tmp);
    free(ops_tmp);
    free(ops_len_tmp);
    free(cons_len);
    return any_fail ? JTK_ERR_CHUNK_FAILED : 0;
}

/* kiley polish_until_converge_antidiagonal(template, seqs, ops, strands, HMMPolishConfig::new(radius, take_num,
 * ignore_edge)) on a batch of independent windows: the call of consensus::polish_seg (consensus/mod.rs:476-483) and of the
 * stage itself (mod.rs:105-106).  radius 0 = ceil(len * band_frac) / 2 (mod.rs:96,105); take_num 0 = all reads. */
int jo_polish_chunks(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                     const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                     const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, uint32_t radius,
                     uint32_t take_num, uint32_t ignore_edge, uint8_t *cons_out, uint64_t *cons_off, uint8_t *ops_out,
                     uint64_t *ops_out_off, jtk_lc_result_t *result, int n_threads) {
    int any_fail = 0;
    uint8_t **cons_tmp = (uint8_t **)calloc(n_chunks, sizeof(uint8_t *));
    uint8_t ***ops_tmp = (uint8_t ***)calloc(n_chunks, sizeof(uint8_t **));
    size_t **ops_len_tmp = (size_t **)calloc(n_chunks, sizeof(size_t *));
    size_t *cons_len = (size_t *)calloc(n_chunks, sizeof(size_t));
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (long ci = 0; ci < (long)n_chunks; ci++) {
        const jtk_lc_chunk_t *ch = &chunks[ci];
        size_t n = ch->n_reads, tl = (size_t)ch->tmpl_len;
        const uint8_t **reads = (const uint8_t **)malloc((n ? n : 1) * sizeof(*reads));
        size_t *rlen = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
        uint8_t **rops = (uint8_t **)malloc((n ? n : 1) * sizeof(*rops));
        size_t *olen = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
        size_t slack = tl / 4 + 64, max_ops = 0;
        for (size_t r = 0; r < n; r++) {
            size_t g = (size_t)ch->read_first + r, l = (size_t)(ops_off[g + 1] - ops_off[g]);
            if (l > max_ops) max_ops = l;
        }
        size_t ocap = max_ops + slack;
        for (size_t r = 0; r < n; r++) {
            size_t g = (size_t)ch->read_first + r;
            reads[r] = read_bases + read_off[g];
            rlen[r] = (size_t)(read_off[g + 1] - read_off[g]);
            olen[r] = (size_t)(ops_off[g + 1] - ops_off[g]);
            rops[r] = (uint8_t *)malloc(ocap + 8);
            memcpy(rops[r], ops + ops_off[g], olen[r]);
        }
        size_t ccap = tl + slack;
        uint8_t *cons = (uint8_t *)malloc(ccap + 8);
        size_t rad = radius ? radius : band_width_of(params->band_frac, tl) / 2;
        uint32_t rounds = 0;
        int64_t cl = jo_phmm_polish(&params->forward, &params->reverse, tmpl_bases + ch->tmpl_off, tl, n, reads, rlen, rops,
                                    olen, ocap, strand + ch->read_first, rad, take_num ? take_num : n, ignore_edge, cons,
                                    ccap, &rounds);
        memset(&result[ci], 0, sizeof result[ci]);
        result[ci].cluster_num = 1;
        result[ci].polish_rounds = rounds;
        result[ci].status = cl < 0 ? JTK_ERR_CHUNK_FAILED : 0;
        if (cl < 0) {
#pragma omp atomic write
            any_fail = 1;
        }
        cons_tmp[ci] = cons;
        cons_len[ci] = cl < 0 ? 0 : (size_t)cl;
        ops_tmp[ci] = rops;
        ops_len_tmp[ci] = olen;
        free(reads);
        free(rlen);
    }
    uint64_t co = 0, oo = 0;
    for (size_t ci = 0; ci < n_chunks; ci++) {
        const jtk_lc_chunk_t *ch = &chunks[ci];
        cons_off[ci] = co;
        memcpy(cons_out + co, cons_tmp[ci], cons_len[ci]);
        co += cons_len[ci];
        for (size_t r = 0; r < ch->n_reads; r++) {
            size_t g = (size_t)ch->read_first + r;
            size_t l = result[ci].status == 0 ? ops_len_tmp[ci][r] : 0;
            ops_out_off[g] = oo;
            memcpy(ops_out + oo, ops_tmp[ci][r], l);
            oo += l;
            ops_out_off[g + 1] = oo;
            free(ops_tmp[ci][r]);
        }
        free(ops_tmp[ci]);
        free(ops_len_tmp[ci]);
        free(cons_tmp[ci]);
    }
    cons_off[n_chunks] = co;
    free(cons_tmp);
    free(ops_tmp);
    free(ops_len_tmp);
    free(cons_len);
    return any_fail ? JTK_ERR_CHUNK_FAILED : 0;
}

int jo_cluster_features(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_feature_chunk_t *chunks,
                        const double *variants, const uint32_t *variant_type, uint32_t *label,
                        double *log_post, uint32_t post_stride, jtk_lc_result_t *result, int n_threads) {
    int any_fail = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (long ci = 0; ci < (long)n_chunks; ci++) {
        const jtk_lc_feature_chunk_t *ch = &chunks[ci];
        size_t n = ch->n_reads, dim = ch->dim, copy_num = ch->copy_num;
        jo_rng_t rng;
        jo_rng_seed_from_u64(&rng, ch->chunk_id * 3490ULL);
        jo_cluster_config_t cfg;
        cfg.band_width = 0;
        cfg.gains = &params->gains;
        cfg.coverage = params->haploid_coverage;
        cfg.copy_num = copy_num;
        cfg.local_coverage = ch->local_coverage;
        size_t *vh = (size_t *)malloc((dim ? dim : 1) * sizeof(size_t));
        int *vt = (int *)malloc((dim ? dim : 1) * sizeof(int));
        for (size_t d = 0; d < dim; d++) {
            vh[d] = variant_type[2 * (ch->vt_off + d)];
            vt[d] = (int)variant_type[2 * (ch->vt_off + d) + 1];
        }
        size_t *assign = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
        size_t kcap = copy_num > 1 ? copy_num : 1;
        double *pk = (double *)malloc((n ? n : 1) * kcap * sizeof(double));
        double score = 0;
        size_t k = 1;
        int rc;
        if (copy_num < 2) { /* pseudo_mcmc.rs:86-88 */
            for (size_t i = 0; i < n; i++) {
                assign[i] = 0;
                pk[i] = 0.0;
            }
            rc = 0;
        } else {
            rc = jo_cluster_filtered_variants(variants + ch->var_off, n, dim, vh, vt, &cfg, &rng, assign, pk,
                                              &score, &k);
            if (rc == 0) jo_reassign_and_posterior(n, k, assign, pk);
        }
        double *post = log_post + (size_t)ch->read_first * post_stride;
        for (size_t i = 0; i < n; i++) {
            label[ch->read_first + i] = rc == 0 ? (uint32_t)assign[i] : 0;
            for (size_t c = 0; c < post_stride; c++) post[i * post_stride + c] = 0.0;
            if (rc == 0)
                for (size_t c = 0; c < k && c < post_stride; c++) post[i * post_stride + c] = pk[i * k + c];
        }
        result[ci].score = score;
        result[ci].cluster_num = (uint32_t)k;
        result[ci].status = rc == 0 ? 0 : JTK_ERR_CHUNK_FAILED;
        result[ci].polish_rounds = 0;
        result[ci].n_variants = (uint32_t)dim;
        if (rc != 0) {
#pragma omp atomic write
            any_fail = 1;
        }
        free(vh);
        free(vt);
        free(assign);
        free(pk);
    }
    return any_fail ? JTK_ERR_CHUNK_FAILED : 0;
}

int jo_modification_table(const jtk_lc_params_t *params, const uint8_t *tmpl, uint64_t tmpl_len,
                          uint32_t n_reads, const uint8_t *read_bases, const uint64_t *read_off,
                          const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, double *table,
                          double *lk) {
    size_t cols = (size_t)JTK_NUM_ROW * ((size_t)tmpl_len + 1);
    size_t radius = band_width_of(params->band_frac, (size_t)tmpl_len) / 2;
#pragma omp parallel for schedule(dynamic, 1)
    for (long r = 0; r < (long)n_reads; r++) {
        const jtk_hmm_t *h = strand[r] ? &params->forward : &params->reverse;
        double l = jo_phmm_modification_table(h, tmpl, (size_t)tmpl_len, read_bases + read_off[r],
                                              (size_t)(read_off[r + 1] - read_off[r]), ops + ops_off[r],
                                              (size_t)(ops_off[r + 1] - ops_off[r]), radius,
                                              table + (size_t)r * cols);
        for (size_t p = 0; p < cols; p++) table[(size_t)r * cols + p] -= l; /* pseudo_mcmc.rs:64 */
        lk[r] = l;
    }
    return 0;
}
