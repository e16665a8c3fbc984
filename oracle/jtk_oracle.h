/* jtk_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the per-chunk local-clustering path of ban-m/jtk
 * (haplotyper/src/local_clustering/{mod,pseudo_mcmc,normalize}.rs, likelihood_gains.rs, misc.rs) used
 * only to CHECK the HIP path: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it; nothing under jtk_amd/ may.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *  - in-tree logic (pseudo_mcmc.rs, misc.rs k-means, normalize.rs, likelihood_gains.rs tables): follows
 *    the Rust line by line; pinned by the reference's four unit tests on this path and the
 *    runtime invariants it asserts.
 *  - rand 0.8.5 / rand_xoshiro 0.6.0 sampling (not in /root/reference): restated from the published
 *    algorithms; pinned by the public xoshiro256** / SplitMix64 known-answer vectors only.
 *  - kiley 0.3.0 @34ebbda (pair-HMM, polishing, read simulation; not in /root/reference, no tests at the
 *    call sites): OWN SPECIFICATION, **parity unpinned** -- phmm.c states the algorithm this build uses.
 *  - f64 exp/ln: include/jtk_math.h (fdlibm) instead of the platform libm, on both sides.
 */
#ifndef JTK_ORACLE_H
#define JTK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "jtk_lc.h" /* POD structs shared with the ABI: jtk_hmm_t, jtk_gains_t, jtk_lc_params_t */

#ifdef __cplusplus
extern "C" {
#endif

#define JO_MASK_LENGTH 7      /* pseudo_mcmc.rs:3 */
#define JO_MAX_HOMOP_LENGTH 2 /* pseudo_mcmc.rs:4 */
#define JO_POS_THR 0.00001    /* pseudo_mcmc.rs:5 */

/* ---------------- rng.c: rand_xoshiro 0.6.0 Xoshiro256StarStar + rand 0.8.5 distributions ---------- */
typedef struct jo_rng {
    uint64_t s[4];
    uint64_t draws; /* number of next_u64 calls so far (test aid) */
    uint64_t kind;  /* 0 = Xoshiro256StarStar (the clustering stage), 1 = Xoroshiro128PlusPlus (correct_clustering) */
} jo_rng_t;
uint64_t jo_splitmix64_next(uint64_t *x);
void jo_rng_seed_from_u64(jo_rng_t *rng, uint64_t seed);
void jo_rng128pp_seed_from_u64(jo_rng_t *rng, uint64_t seed); /* Xoroshiro128PlusPlus::seed_from_u64 */
uint64_t jo_rng_next_u64(jo_rng_t *rng);
uint32_t jo_rng_next_u32(jo_rng_t *rng);
uint64_t jo_gen_range_usize(jo_rng_t *rng, uint64_t n); /* rng.gen_range(0..n), n: usize  */
uint32_t jo_gen_range_u32(jo_rng_t *rng, uint32_t n);   /* rng.gen_range(0..n), n: u32    */
uint64_t jo_gen_index(jo_rng_t *rng, uint64_t ubound);  /* rand::seq::gen_index           */
int jo_gen_bool(jo_rng_t *rng, double p);               /* rng.gen_bool(p)                */
/* (0..k).filter(|&c| c != old).choose(rng).unwrap()  (pseudo_mcmc.rs:732) */
uint64_t jo_choose_other(jo_rng_t *rng, uint64_t k, uint64_t old);
/* slice.choose_weighted(rng, |i| w[i]).unwrap(): index, or -1 on WeightedError (the reference panics) */
int64_t jo_choose_weighted(jo_rng_t *rng, const double *w, size_t n);

/* ---------------- misc.c ---------------------------------------------------------------------------- */
double jo_logsumexp(const double *xs, size_t n);                          /* misc.rs:84-92   */
double jo_rand_index(const size_t *label, const size_t *pred, size_t n);  /* misc.rs:5-20    */
/* misc.rs:231-259; data is n x dim row-major; returns 0 or -1 (reference panic) */
int jo_kmeans(const double *data, size_t n, size_t dim, size_t k, jo_rng_t *rng, double *dist_out,
              size_t *assign);
/* kiley_op_to_ops (misc.rs:188-225): per-base ops -> run-length (kind 0=M,1=D,2=I ; len). returns #runs */
size_t jo_ops_to_runs(const uint8_t *ops, size_t n, uint8_t *kind, uint64_t *len);
/* Node::recover based sort key (mod.rs:47-50) */
uint64_t jo_pileup_sort_key(const uint8_t *tmpl, size_t tl, const uint8_t *read, size_t rl,
                            const uint8_t *ops, size_t n_ops);

/* ---------------- likelihood_gains.c ---------------------------------------------------------------- */
double jo_gains_expected(const jtk_gains_t *g, size_t homop_len, int diff_type);  /* :79-87  */
/* pvalues(prob, n) (:115-129): out has n+1 entries */
void jo_pvalues(double prob, size_t n, double *out);
/* estimate_gain (:162-184) with the own-spec read simulator / bootstrap likelihood of phmm.c */
void jo_estimate_gain(const jtk_hmm_t *fwd, const jtk_hmm_t *rev, uint64_t seed, size_t seq_len,
                      size_t band, size_t homop_len, jtk_gains_t *out);
void jo_estimate_gain_default(const jtk_hmm_t *fwd, const jtk_hmm_t *rev, jtk_gains_t *out); /* :190 */
/* likelihood_gains.rs:6-39 (reference constants: seed 23908, 1000 samples, 500 reads, length 100, band 25) */
double jo_estimate_minimum_gain(const jtk_hmm_t *fwd, const jtk_hmm_t *rev, uint64_t seed, size_t sample_num, size_t seq_num,
                                size_t len, size_t band, int n_threads);

/* ---------------- pseudo_mcmc.c --------------------------------------------------------------------- */
typedef struct jo_cluster_config { /* pseudo_mcmc.rs:17-25 */
    size_t band_width;
    const jtk_gains_t *gains;
    double coverage;
    size_t copy_num;
    double local_coverage;
} jo_cluster_config_t;

/* the reference's trace! rows of a chunk's clustering (TOTAL / CAND / PICK / DUMP / RANGE / LK / COUNTS; pseudo_mcmc.c says which
 * lines) go to the sink while one is set: '\n'-terminated rows; len keeps counting beyond cap */
typedef struct jo_trace {
    char *text;
    size_t cap, len;
} jo_trace_t;
void jo_trace_set(jo_trace_t *t);
void jo_homopolymer_length(const uint8_t *xs, size_t n, size_t *out);        /* :195-211 */
double jo_cosine_similarity(const double *profiles, size_t n, size_t cols, size_t i, size_t j); /* :602 */
double jo_sokal_michener(const double *profiles, size_t n, size_t cols, size_t i, size_t j);    /* :618 */
double jo_poisson_lk(size_t x, double lambda);                               /* :636-638 */
double jo_max_poisson_lk(size_t x, double lambda, size_t c_start, size_t c_end); /* :641-645 */
void jo_compress_small_gains(double *profiles, size_t n, size_t cols, const uint8_t *tmpl, size_t tl,
                             const jtk_gains_t *gains);                      /* :141-165 */
/* filter_profiles (:426-474): returns number of probes D; pos_out/score_out need capacity
 * 3*max(copy_num,2) */
size_t jo_filter_profiles(const uint8_t *tmpl, size_t tl, const double *profiles, size_t n,
                          const uint8_t *strands, const jo_cluster_config_t *cfg, size_t *pos_out,
                          double *score_out);
/* mcmc_with_filter (:704-762) */
double jo_mcmc_with_filter(const double *data, size_t n, size_t dim, size_t *assign, size_t k, double cov,
                           jo_rng_t *rng);
/* mcmc_clustering (:649-670); returns 0 / -1 */
int jo_mcmc_clustering(const double *data, size_t n, size_t dim, size_t k, double cov, jo_rng_t *rng,
                       size_t *assign, double *score, double *lk_gains, uint8_t *used_columns);
/* cluster_filtered_variants (:213-274); likelihood_gains is n x max_k (caller gives n x copy_num).
 * returns 0 / -1 (reference would panic) */
int jo_cluster_filtered_variants(const double *variants, size_t n, size_t dim, const size_t *vt_homop,
                                 const int *vt_type, const jo_cluster_config_t *cfg, jo_rng_t *rng,
                                 size_t *assign, double *likelihood_gains, double *score, size_t *k_out);
/* tail of clustering() (:98-105): arg-max re-assignment + to_posterior_probability, in place */
void jo_reassign_and_posterior(size_t n, size_t k, size_t *assign, double *likelihood_gains);
/* clustering() (:77-107) on an (already polished) template; post is n x copy_num capacity */
int jo_clustering(const uint8_t *tmpl, size_t tl, size_t n, const uint8_t *const *reads,
                  const size_t *read_len, const uint8_t *const *ops, const size_t *ops_len,
                  const uint8_t *strands, jo_rng_t *rng, const jtk_hmm_t *fwd, const jtk_hmm_t *rev,
                  const jo_cluster_config_t *cfg, size_t *assign, double *post, double *score,
                  size_t *k_out, size_t *n_variants_out);
/* search_variants (:109-138): variants (n x D) + types; returns D. Buffers need 3*max(copy_num,2). */
size_t jo_search_variants(const uint8_t *tmpl, size_t tl, size_t n, const uint8_t *const *reads,
                          const size_t *read_len, const uint8_t *const *ops, const size_t *ops_len,
                          const uint8_t *strands, const jtk_hmm_t *fwd, const jtk_hmm_t *rev,
                          const jo_cluster_config_t *cfg, double *variants, size_t *vt_homop, int *vt_type,
                          size_t *pos_out);

/* ---------------- exact_clustering.c: exact_clustering.rs:7-26 (the comparator of benchmark_mcmc.rs) -------- */
double jo_cluster_filtered_variants_exact(const double *variants, size_t n, size_t dim, size_t copy_num,
                                          size_t *assign, double *lk_gain);

/* ---------------- normalize.c ----------------------------------------------------------------------- */
void jo_reorder_f64(double *xs, uint64_t *indices, size_t n); /* normalize.rs:54-63 */
void jo_reorder_i64(int64_t *xs, uint64_t *indices, size_t n);
void jo_normalize_pileup(size_t n, size_t cluster_num, uint64_t *cluster, double *post, size_t stride);

/* ---------------- phmm.c: OWN SPEC of the kiley pair-HMM (parity unpinned) -------------------------- */
#define JO_LOG_ZERO (-1.0e300)
#define JO_SCALE_BLOCK 64
/* band centres c[0..tl+rl] from ops; returns 0 or -1 if ops do not consume template and read exactly */
int jo_band_centers(const uint8_t *ops, size_t n_ops, size_t tl, size_t rl, uint32_t *centers);
/* log P(read | template) in the ops-guided band (kiley likelihood_antidiagonal given ops) */
double jo_phmm_likelihood(const jtk_hmm_t *hmm, const uint8_t *tmpl, size_t tl, const uint8_t *read,
                          size_t rl, const uint8_t *ops, size_t n_ops, size_t radius);
/* kiley modification_table_antidiagonal: table has JTK_NUM_ROW*(tl+1) log-likelihoods (NOT minus lk);
 * returns lk */
double jo_phmm_modification_table(const jtk_hmm_t *hmm, const uint8_t *tmpl, size_t tl, const uint8_t *read,
                                  size_t rl, const uint8_t *ops, size_t n_ops, size_t radius,
                                  double *table);
/* kiley polish_until_converge_antidiagonal(template, seqs, ops, strands, HMMPolishConfig{radius,
 * take_num, ignore_edge}).  ops[r] are updated in place (capacity ops_cap each, lengths in ops_len);
 * consensus written to cons (capacity cons_cap), length returned (or -1). rounds_out optional. */
int64_t jo_phmm_polish(const jtk_hmm_t *fwd, const jtk_hmm_t *rev, const uint8_t *tmpl, size_t tl, size_t n,
                       const uint8_t *const *reads, const size_t *read_len, uint8_t **ops, size_t *ops_len,
                       size_t ops_cap, const uint8_t *strands, size_t radius, size_t take_num,
                       size_t ignore_edge, uint8_t *cons, size_t cons_cap, uint32_t *rounds_out);
/* global unit-cost alignment ops (stand-in for the bootstrap alignment of kiley *_bootstrap functions
 * and for minimap2+edlib cigars in the synthetic data); returns n_ops, ops capacity tl+rl */
size_t jo_edit_ops(const uint8_t *tmpl, size_t tl, const uint8_t *read, size_t rl, uint8_t *ops);
/* kiley likelihood_antidiagonal_bootstrap */
double jo_phmm_likelihood_bootstrap(const jtk_hmm_t *hmm, const uint8_t *tmpl, size_t tl,
                                    const uint8_t *read, size_t rl, size_t radius);
/* kiley gen_seq::generate_seq and hmm Generate::gen (own spec); out capacity 3*tl+16; returns length */
void jo_generate_seq(jo_rng_t *rng, size_t len, uint8_t *out);
size_t jo_phmm_gen(const jtk_hmm_t *hmm, const uint8_t *tmpl, size_t tl, jo_rng_t *rng, uint8_t *out,
                   size_t cap);

/* ---------------- model_fit.c: model_tune.rs:96-156 (driver in tree; the Baum-Welch step is OWN SPEC, kiley absent) ---- */
#define JTK_FIT_COUNTS 45 /* 9 transitions (row-major M,I,D x M,I,D) + mat_emit[16] + ins_emit[20] */
double jo_phmm_counts(const jtk_hmm_t *hmm, const uint8_t *tmpl, size_t tl, const uint8_t *read, size_t rl,
                      const uint8_t *ops, size_t n_ops, size_t radius, double *counts);
void jo_fit_mstep(const jtk_hmm_t *old, const double *counts, jtk_hmm_t *out);
int jo_fit_model(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks, const uint8_t *tmpl_bases,
                 const uint8_t *read_bases, const uint64_t *read_off, const uint8_t *ops, const uint64_t *ops_off,
                 const uint8_t *strand, uint32_t rounds, jtk_hmm_t *fwd_out, jtk_hmm_t *rev_out);

/* ---------------- correction.c: phmm_likelihood_correction.rs:32-97 ------------------------------------------------ */
/* ari_out[n_chunks] (optional): the adjusted Rand index of every corrected chunk; sims_first (optional): the similarity
 * matrix of the first corrected chunk, members in correct_chunk's order (test aids) */
int jo_correct_clustering(size_t n_reads, const uint64_t *read_id, const uint64_t *node_off, const jtk_cc_node_t *nodes,
                          const double *posteriors, size_t n_chunks, jtk_cc_chunk_t *chunks, size_t n_selected,
                          const uint64_t *selection, double haploid_coverage, double min_gain, uint64_t *cluster_out,
                          uint8_t *touched, double *ari_out, double *sims_first);

/* ---------------- local_clustering.c: mod.rs ------------------------------------------------------- */
typedef struct jo_chunk_result {
    double score;
    size_t cluster_num;
    size_t cons_len;
    uint32_t polish_rounds;
    size_t n_variants;
    double elapsed_ms, polish_ms;
} jo_chunk_result_t;
/* clustering_on_pileup (mod.rs:86-123), copy_num < 8.  ops updated in place (capacity ops_cap).
 * post is n x post_stride. skip_polish != 0 == jtk_lc_cluster_polished semantics. returns 0 / <0 */
int jo_clustering_on_pileup(const jtk_lc_params_t *params, uint64_t chunk_id, size_t copy_num,
                            const uint8_t *tmpl, size_t tl, size_t n, const uint8_t *const *reads,
                            const size_t *read_len, uint8_t **ops, size_t *ops_len, size_t ops_cap,
                            const uint8_t *strands, int skip_polish, size_t *assign, double *post,
                            size_t post_stride, uint8_t *cons, size_t cons_cap, jo_chunk_result_t *res);
/* Flat-array batch driver with the same signature family as jtk_lc_cluster_chunks (OpenMP over chunks
 * mirrors the rayon loop of mod.rs:64-72); n_threads <= 0 means all cores. */
int jo_cluster_chunks(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                      const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                      const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, int skip_polish,
                      uint32_t *label, double *log_post, uint32_t post_stride, jtk_lc_result_t *result,
                      uint8_t *cons_out, uint64_t *cons_off, uint64_t cons_cap, uint8_t *ops_out,
                      uint64_t *ops_out_off, uint64_t ops_cap, int n_threads, double *record_ms);
int jo_polish_chunks(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                     const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                     const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, uint32_t radius,
                     uint32_t take_num, uint32_t ignore_edge, uint8_t *cons_out, uint64_t *cons_off, uint8_t *ops_out,
                     uint64_t *ops_out_off, jtk_lc_result_t *result, int n_threads);
int jo_cluster_features(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_feature_chunk_t *chunks,
                        const double *variants, const uint32_t *variant_type, uint32_t *label,
                        double *log_post, uint32_t post_stride, jtk_lc_result_t *result, int n_threads);
int jo_modification_table(const jtk_lc_params_t *params, const uint8_t *tmpl, uint64_t tmpl_len,
                          uint32_t n_reads, const uint8_t *read_bases, const uint64_t *read_off,
                          const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, double *table,
                          double *lk);
double jo_exp(double x);
double jo_log(double x);

/* test hooks (correction.c): include/jtk_eigen.h and misc.rs adjusted_rand_index :22-46 as the correction uses them */
void jo_symmetric_eigen(double *a, size_t n, double *v);
double jo_adjusted_rand_index(const size_t *label, const size_t *pred, size_t n);

#ifdef __cplusplus
}
#endif
#endif
