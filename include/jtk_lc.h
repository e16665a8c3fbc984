/* jtk_lc.h -- C ABI of the MI355X-native local-clustering stage (drop-in for the rayon loop of
 * ban-m/jtk haplotyper/src/local_clustering/mod.rs:64-72).
 *
 * Everything here is plain C: pointers, sizes and POD structs; no torch / HIP types.  All buffers are
 * caller-owned HOST memory unless a function name ends in `_dev`; the library never keeps a pointer
 * after returning, never throws/unwinds across the boundary, and returns 0 or a negative jtk_status.
 * The reference has no FFI layer (it is one Rust process); each entry point below cites the Rust item
 * it replaces, and INTEGRATION.md shows the `extern "C"` block + shim a jtk maintainer would add.
 *
 * Sequence encoding at the boundary: ASCII upper-case ACGT (the reference rejects anything else at
 * entry, haplotyper/src/entry.rs:40-45).  Alignment ops are one byte per column, values of
 * `kiley::Op` in the order the reference uses them (haplotyper/src/misc.rs:167-172):
 *   0 = Match, 1 = Mismatch, 2 = Ins (read base, no template base), 3 = Del (template base, no read base).
 */
#ifndef JTK_LC_H
#define JTK_LC_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: the entry points below (and the diagnostic ones of jtk_lc_debug.h) are
 * its only dynamic symbols. */
#if defined(__GNUC__) || defined(__clang__)
#define JTK_LC_API __attribute__((visibility("default")))
#else
#define JTK_LC_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a public struct changes size or an entry point gains a precondition; a host compares it with
 * jtk_lc_version() before the first call.  2: jtk_lc_timing_t grew by chain_lds_bytes[2] (round 3) and
 * jtk_lc_cluster_chunks_multi checks the contiguous read_first layout itself (see there). */
#define JTK_LC_ABI_VERSION 2

/* kiley::hmm::NUM_ROW = 8 + COPY_SIZE + DEL_SIZE (pseudo_mcmc.rs:7,172,447): rows 0-3 substitute to
 * ACGT, 4-7 insert ACGT before the position, 8-10 copy 1-3 bp, 11-13 delete 1-3 bp. */
#define JTK_NUM_ROW 14
#define JTK_COPY_SIZE 3
#define JTK_DEL_SIZE 3
#define JTK_GAINS_MAX_HOMOP 8

enum jtk_op { JTK_OP_MATCH = 0, JTK_OP_MISMATCH = 1, JTK_OP_INS = 2, JTK_OP_DEL = 3 };

/* likelihood_gains.rs:194-199 (declaration order). */
enum jtk_diff_type { JTK_DIFF_SUBST = 0, JTK_DIFF_DEL = 1, JTK_DIFF_INS = 2 };

typedef enum jtk_status {
    JTK_OK = 0,
    JTK_ERR_INVALID_ARG = -1,     /* null pointer, inconsistent offsets, non-ACGT base, bad op code    */
    JTK_ERR_NO_DEVICE = -2,       /* no usable MI355X / HIP runtime failure (message via last_error)   */
    JTK_ERR_UNSUPPORTED = -3,     /* band radius > 255 (any read count is taken)                        */
    JTK_ERR_ALLOC = -4,           /* hipMalloc / host allocation failed                                */
    JTK_ERR_OPS_MISMATCH = -5,    /* ops do not consume exactly the template and the read              */
    JTK_ERR_CHUNK_FAILED = -6,    /* >=1 chunk hit a condition on which the reference panics; see      */
                                  /* chunk_status[] (e.g. WeightedError::AllWeightsZero, misc.rs:335)  */
    JTK_ERR_INTERNAL = -7
} jtk_status;

/* definitions/src/lib.rs:101-126 (same field order; model_tune.rs:36-63 maps it 1:1 onto
 * kiley::hmm::PairHiddenMarkovModel).  mat_emit[4*ref + read]; ins_emit[4*prev_read_base + read] with
 * prev = 4 for the first read base (own reading of kiley's 20-entry table). */
typedef struct jtk_hmm {
    double mat_mat, mat_ins, mat_del;
    double ins_mat, ins_ins, ins_del;
    double del_mat, del_ins, del_del;
    double mat_emit[16];
    double ins_emit[20];
} jtk_hmm_t;

/* likelihood_gains.rs:41-47 / :55-61. */
typedef struct jtk_gain_profile {
    double gain;
    double prob;
} jtk_gain_profile_t;

typedef struct jtk_gains {
    uint32_t max_homopolymer_len; /* 3 for estimate_gain_default (likelihood_gains.rs:189) */
    uint32_t reserved;
    jtk_gain_profile_t subst[JTK_GAINS_MAX_HOMOP];
    jtk_gain_profile_t deletions[JTK_GAINS_MAX_HOMOP];
    jtk_gain_profile_t insertions[JTK_GAINS_MAX_HOMOP];
} jtk_gains_t;

/* What local_clustering_selected computes once per call and passes by reference into every
 * clustering_on_pileup (mod.rs:57-62): the model on both strands, the calibrated gains, the haploid
 * coverage and the read type (only its band fraction matters here, definitions/src/lib.rs:173-175). */
typedef struct jtk_lc_params {
    jtk_hmm_t forward;
    jtk_hmm_t reverse;
    jtk_gains_t gains;
    double haploid_coverage;
    double band_frac; /* 0.03 ONT, 0.01 CCS, 0.05 CLR/None; band_width = ceil(len*frac), radius = band_width/2 */
} jtk_lc_params_t;

/* One pile-up = one call of clustering_on_pileup (mod.rs:86-123). Reads are already in the order fixed
 * by pileup_nodes (mod.rs:45-51); jtk_lc_pileup_sort_keys below gives the sort key. */
typedef struct jtk_lc_chunk {
    uint64_t chunk_id;   /* seeds the per-chunk RNG: Xoshiro256StarStar::seed_from_u64(id * 3490) (mod.rs:97) */
    uint32_t copy_num;   /* Chunk.copy_num (definitions/src/lib.rs:403-415) */
    uint32_t n_reads;
    uint64_t tmpl_off;   /* into tmpl_bases */
    uint64_t tmpl_len;
    uint64_t read_first; /* index of this chunk's first read in read_off / ops_off / strand */
} jtk_lc_chunk_t;

/* Per-chunk result record. */
typedef struct jtk_lc_result {
    double score;         /* Chunk.score (mod.rs:78)        */
    uint32_t cluster_num; /* Chunk.cluster_num (mod.rs:79)   */
    int32_t status;       /* 0, or the jtk_status this chunk failed with */
    uint32_t polish_rounds;
    uint32_t n_variants;  /* D: number of selected variant columns (filter_profiles) */
} jtk_lc_result_t;

/* ---- the drop-in entry point ------------------------------------------------------------------
 * Replaces `pileups.into_par_iter()...map(clustering_on_pileup)` (mod.rs:64-72) for a batch of chunks:
 * consensus polishing (mod.rs:105-106), variant search (pseudo_mcmc.rs:109-138), clustering
 * (pseudo_mcmc.rs:213-274, 77-107) and the per-node write-back data of update_by_clusterings
 * (mod.rs:244-260).  Chunks with copy_num >= 8 take clustering_recursive's split branch (mod.rs:138-189):
 * a 4-way clustering, then per group a consensus polish and a clustering with the group's share of the
 * copies, merged as mod.rs:161-187 does; post_stride must be >= the largest copy_num of the batch.
 *
 * label[r]        : Node.cluster of read r (before normalize_local_clustering).
 * log_post        : row r has post_stride doubles; the first result[c].cluster_num are Node.posterior
 *                   (log posterior, rows logsumexp to 0), the rest are written as 0.
 * cons_out        : polished consensus of chunk c at cons_off[c] .. cons_off[c+1]; the caller provides
 *                   capacity cons_cap bytes in total (>= sum(tmpl_len) + 64*n_chunks is always enough:
 *                   the library fails the chunk rather than overrun).
 * ops_out         : re-threaded per-base ops of read r at ops_out_off[r] .. ops_out_off[r+1], capacity
 *                   ops_cap bytes (sum(ops_len) + 8*polish edits; same overrun rule).
 * device          : HIP device ordinal (one process drives one GPU; multi-GPU = one process per GPU).
 */
JTK_LC_API int jtk_lc_cluster_chunks(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                          const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                          const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand,
                          uint32_t *label, double *log_post, uint32_t post_stride, jtk_lc_result_t *result,
                          uint8_t *cons_out, uint64_t *cons_off, uint64_t cons_cap, uint8_t *ops_out,
                          uint64_t *ops_out_off, uint64_t ops_cap, int device);

/* The same call over several GPUs of one node from one host process (SURVEY 8b's `device_mask`, as a list): chunks are
 * dealt to `devices` by longest-processing-time-first over a cost model (pair-HMM cells + Metropolis steps per candidate k;
 * the partition bench.py / jtk_amd.sharding use between ranks), each share is gathered into its own batch and runs as on a
 * single device (sliced, overlapped), and the outputs are scattered back into the caller's order, laid out as above.  Chunks are independent (one RNG stream per chunk id), so results do not depend on
 * the device list; the path has no exchange step and no collective runs.  A device may be listed more than once.
 * Precondition (checked here since ABI 2, as the single-device call always did when it built its session): the chunks' reads
 * are laid out back to back in chunk order (chunks[0].read_first == 0, chunks[c + 1].read_first == chunks[c].read_first +
 * chunks[c].n_reads) -- JTK_ERR_INVALID_ARG otherwise. */
JTK_LC_API int jtk_lc_cluster_chunks_multi(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                                const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                                const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand,
                                uint32_t *label, double *log_post, uint32_t post_stride, jtk_lc_result_t *result,
                                uint8_t *cons_out, uint64_t *cons_off, uint64_t cons_cap, uint8_t *ops_out,
                                uint64_t *ops_out_off, uint64_t ops_cap, const int *devices, size_t n_devices);

/* Same, but the template is an already polished consensus and ops are already re-threaded: skips
 * polish_until_converge_antidiagonal.  This is `pseudo_mcmc::clustering` (pseudo_mcmc.rs:77-107) batched;
 * a Rust host that keeps real kiley polishing calls this one and inherits exactness downstream.
 * cons_out/ops_out are not produced. */
JTK_LC_API int jtk_lc_cluster_polished(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                            const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                            const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand,
                            uint32_t *label, double *log_post, uint32_t post_stride,
                            jtk_lc_result_t *result, int device);

/* ---- the polishing step on its own -------------------------------------------------------------------
 * kiley `polish_until_converge_antidiagonal(template, seqs, ops, strands, &HMMPolishConfig::new(radius, take_num,
 * ignore_edge))` for a batch of independent windows (one jtk_lc_chunk_t each; copy_num is ignored): the call
 * `consensus::polish_seg` makes on 2 kbp windows of contigs (haplotyper/src/consensus/mod.rs:476-483 with
 * (radius / 2, max_coverage, 0)) and the one this stage makes per pile-up (local_clustering/mod.rs:105-106 with
 * (band / 2, N, 3)).  Only the first take_num reads of a window vote in a round (0 = all); every read's ops are
 * re-threaded.  radius 0 derives the radius from the window length and params->band_frac as mod.rs:96 does.
 * result[c].polish_rounds / .status are filled; cons_out / ops_out as in jtk_lc_cluster_chunks. */
JTK_LC_API int jtk_lc_polish_chunks(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                         const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                         const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, uint32_t radius,
                         uint32_t take_num, uint32_t ignore_edge, uint8_t *cons_out, uint64_t *cons_off, uint64_t cons_cap,
                         uint8_t *ops_out, uint64_t *ops_out_off, uint64_t ops_cap, jtk_lc_result_t *result, int device);

/* ---- resident-batch form of the same call ----------------------------------------------------------
 * jtk_lc_cluster_chunks == session_create + session_run(0) + session_fetch + session_destroy.
 * A session uploads the batch once (inputs stay resident in HBM, all device workspaces are allocated
 * up front), so repeated runs measure the device path without PCIe; bench.py times session_run.
 * The reference enters this stage up to ~6 times per pipeline run on overlapping chunk sets
 * (cli/src/pipeline.rs:158,164-168), which is what a resident session serves. */
typedef struct jtk_lc_session jtk_lc_session_t;
JTK_LC_API int jtk_lc_session_create(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                          const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                          const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand,
                          uint32_t post_stride, int device, jtk_lc_session_t **out);
/* One pass of the hot path over the resident batch; skip_polish != 0 gives jtk_lc_cluster_polished
 * semantics.  Returns after the device has finished; results stay on the device until fetched. */
JTK_LC_API int jtk_lc_session_run(jtk_lc_session_t *s, int skip_polish);
/* Copies the last run's results out (any pointer may be NULL to skip that output). */
JTK_LC_API int jtk_lc_session_fetch(jtk_lc_session_t *s, uint32_t *label, double *log_post, jtk_lc_result_t *result,
                         uint8_t *cons_out, uint64_t *cons_off, uint64_t cons_cap, uint8_t *ops_out,
                         uint64_t *ops_out_off, uint64_t ops_cap);
JTK_LC_API int jtk_lc_session_destroy(jtk_lc_session_t *s);
/* The reference's trace! rows (log level Trace) of ONE chunk's clustering, after a jtk_lc_session_run: '\n'-terminated rows, in the
 * order the reference logs them -- TOTAL, CAND per candidate column (pseudo_mcmc.rs:467-472), PICK per picked column (:539),
 * DUMP per selected column (:122-127), RANGE (:236), and per candidate cluster count the two LK rows (:250, :256) and, where the
 * count was accepted, COUNTS (:262).  (Not written: the per-column PVALUE / RAWCOUNT / FILTER rows of the filter, REMOVE, VARS.)
 * The chunk's pick and its chain run once more on the device, in instantiations that record what the rows need (the product
 * kernels carry none of it); labels, posteriors and scores come out as they were.  *len = bytes of text (no terminator);
 * JTK_ERR_INVALID_ARG with *len set when cap is too small; JTK_ERR_UNSUPPORTED for a session that holds a chunk of copy number
 * >= 8 (clustering_recursive, mod.rs:125-189); a chunk of copy number < 2 logs nothing (pseudo_mcmc.rs:86-88). */
JTK_LC_API int jtk_lc_session_trace(jtk_lc_session_t *s, size_t chunk, char *text, size_t cap, size_t *len);

/* ---- stage pieces, exported because the reference exposes them too ------------------------------ */

/* `pseudo_mcmc::modification_table` (pseudo_mcmc.rs:45-68) for one pile-up: for read r, table[r] has
 * JTK_NUM_ROW*(tmpl_len+1) doubles = kiley modification_table_antidiagonal(...) MINUS lk[r]
 * (pseudo_mcmc.rs:62-64); lk[r] is the read's log-likelihood under the unedited template. */
JTK_LC_API int jtk_lc_modification_table(const jtk_lc_params_t *params, const uint8_t *tmpl, uint64_t tmpl_len,
                              uint32_t n_reads, const uint8_t *read_bases, const uint64_t *read_off,
                              const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand,
                              double *table, double *lk, int device);

/* `pseudo_mcmc::cluster_filtered_variants` + the re-assignment / posterior tail of `clustering`
 * (pseudo_mcmc.rs:213-274, 98-105) on caller-supplied feature matrices (the entry the reference's
 * sandbox/src/bin/benchmark_mcmc.rs:111-114 drives).  Chunk c: variants[var_off[c] ..] is n_reads x dim
 * row-major, variant_type[vt_off..] is dim x (homop_len, jtk_diff_type) pairs of uint32. */
typedef struct jtk_lc_feature_chunk {
    uint64_t chunk_id;
    uint32_t copy_num;
    uint32_t n_reads;
    uint32_t dim;
    uint32_t reserved;
    uint64_t var_off;  /* in doubles */
    uint64_t vt_off;   /* in (uint32,uint32) pairs */
    uint64_t read_first;
    double local_coverage; /* ClusteringConfig.local_coverage (mod.rs:108-112) */
} jtk_lc_feature_chunk_t;

JTK_LC_API int jtk_lc_cluster_features(const jtk_lc_params_t *params, size_t n_chunks,
                            const jtk_lc_feature_chunk_t *chunks, const double *variants,
                            const uint32_t *variant_type, uint32_t *label, double *log_post,
                            uint32_t post_stride, jtk_lc_result_t *result, int device);

/* ---- host-side helpers (no GPU needed) ------------------------------------------------------------ */

/* ---- stage preamble: gains calibration ------------------------------------------------------------
 * Replaces `estimate_gain(hmm, seed, seq_len, band, homop_len)` (likelihood_gains.rs:162-184); the stage
 * calls it through `estimate_gain_default` = (hmm, 309423, 100, 10, 3) (likelihood_gains.rs:186-192,
 * local_clustering/mod.rs:60).  Per (type, homopolymer length) profile: 100 simulated variant pairs x 100
 * simulated reads, every read scored against both members of its pair with the banded pair-HMM
 * (likelihood_antidiagonal_bootstrap); the sampling runs on the host, the bootstrap alignments and the
 * likelihoods on the device.  `out` is what jtk_lc_params_t.gains expects. */
JTK_LC_API int jtk_lc_estimate_gains(const jtk_hmm_t *forward, const jtk_hmm_t *reverse, uint64_t seed, uint32_t seq_len,
                          uint32_t band, uint32_t homop_len, jtk_gains_t *out, int device);

/* `estimate_minimum_gain(&hmm)` (likelihood_gains.rs:6-39), the scale of correct_clustering's protection rule
 * (phmm_likelihood_correction.rs:118: min_gain = estimate_minimum_gain(&hmm) * PROTECT_FACTOR).  sample_num templates of
 * `len` random bases, each with one base deleted (kiley introduce_errors(.., 0, 1, 0)); per template the median over seq_num
 * simulated reads of lk(read | template) - lk(read | template minus the base) with the banded bootstrap likelihood; the
 * third smallest median, at least 1.  The reference's constants: seed 23908, 1000, 500, 100, band 25.  Sampling on the
 * host, alignments and likelihoods on the device (the machinery of jtk_lc_estimate_gains; own specification of kiley). */
JTK_LC_API int jtk_lc_estimate_minimum_gain(const jtk_hmm_t *forward, const jtk_hmm_t *reverse, uint64_t seed, uint32_t sample_num,
                                 uint32_t seq_num, uint32_t len, uint32_t band, double *out, int device);

/* ---- stage preamble: model refit ----------------------------------------------------------------------
 * Replaces `estimate_model_parameters_on_both_strands` (haplotyper/src/model_tune.rs:119-152; entered through
 * `update_models_on_both_strands`, local_clustering/mod.rs:58) on the training pile-ups the host has selected
 * (model_tune.rs:99-118: coverage within 2 of the median, sorted by chunk id, the first TRAIN_UNIT_SIZE = 5):
 * `rounds` (TRAIN_ROUND = 10) times [ polish every pile-up with HMMPolishConfig::new(band / 2, N, 0), then one
 * Baum-Welch step over all of them with the largest band's radius ].  params->forward / reverse are the starting model
 * (DataSet.model_param), params->band_frac the read type's; the refitted models are what jtk_lc_params_t.forward /
 * .reverse then carry.  kiley's fit is not part of the reference tree: the step is this build's own specification
 * (expected transition / emission counts of the banded pair-HMM, rows renormalised). */
JTK_LC_API int jtk_lc_fit_model(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                     const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                     const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, uint32_t rounds,
                     jtk_hmm_t *forward_out, jtk_hmm_t *reverse_out, int device);

/* ---- the first consumer of the stage's posteriors: cross-chunk correction ------------------------------------
 * Replaces `AlignmentCorrection::correct_clustering_selected` (haplotyper/src/phmm_likelihood_correction.rs:32-97): per
 * selected chunk with more than one cluster, a read-by-read similarity matrix from a 3-state affine-gap alignment of the
 * neighbouring nodes' posteriors (`alignment` :466-479, `align_swg` :482-531, `sim` :534-550; on the device), then spectral
 * clustering on the host (graph Laplacian :385-402, eigenvectors below 0.2 :405-464, 20 x misc::kmeans :295-302), the
 * adjusted Rand index against the previous labels (:220-240), suppression of the lowest 5 % (:100-105) unless the chunk's
 * local-clustering score protects it (:108-129).
 * The data set reaches the call flattened: read r owns nodes node_off[r] .. node_off[r+1] (in read order), node e has
 * post_len log-posteriors at posteriors[post_off].  chunks[] is DataSet.selected_chunks (cluster_num is updated in place).
 * min_gain is `estimate_minimum_gain(&hmm) * PROTECT_FACTOR` (:118): a simulation through kiley that the caller runs.
 * Output: cluster_out[e] = Node.cluster after the call, touched[e] = 1 where the reference rewrites the node; its posterior
 * then is -10000 everywhere except 0 at the new cluster, with the chunk's NEW cluster_num entries (:88-93).
 * Returns JTK_ERR_CHUNK_FAILED where the reference panics (a pile-up smaller than 4 x copy_num :348+:361, no eigenvalue
 * below the threshold :455, posterior lengths that differ from cluster_num :547, ...); nothing is written then. */
typedef struct jtk_cc_node {
    uint64_t chunk;      /* Node.chunk                 (definitions/src/lib.rs:672-683) */
    uint64_t cluster;    /* Node.cluster                */
    uint32_t is_forward; /* Node.is_forward             */
    uint32_t post_len;   /* Node.posterior.len()        */
    uint64_t post_off;   /* into `posteriors`           */
} jtk_cc_node_t;
typedef struct jtk_cc_chunk {
    uint64_t id;          /* Chunk.id                   (definitions/src/lib.rs:403-415) */
    uint32_t cluster_num; /* Chunk.cluster_num (in / out) */
    uint32_t copy_num;    /* Chunk.copy_num             */
    double score;         /* Chunk.score                */
} jtk_cc_chunk_t;
JTK_LC_API int jtk_lc_correct_clustering(size_t n_reads, const uint64_t *read_id, const uint64_t *node_off, const jtk_cc_node_t *nodes,
                              const double *posteriors, size_t n_chunks, jtk_cc_chunk_t *chunks, size_t n_selected,
                              const uint64_t *selection, double haploid_coverage, double min_gain, uint64_t *cluster_out,
                              uint8_t *touched, int device);

/* Sort key of pileup_nodes (mod.rs:47-50): number of alignment columns that are not '|' in
 * Node::recover (definitions/src/lib.rs:773-813) for run-length cigar ops given per base. */
JTK_LC_API int jtk_lc_pileup_sort_key(const uint8_t *tmpl, uint64_t tmpl_len, const uint8_t *read, uint64_t read_len,
                           const uint8_t *ops, uint64_t ops_len, uint64_t *key_out);

/* normalize_local_clustering for one pile-up (normalize.rs:21-50): relabels by descending cluster size
 * (ties: larger old index first) and permutes each posterior row in place. */
JTK_LC_API int jtk_lc_normalize_pileup(uint32_t n_reads, uint32_t cluster_num, uint32_t *label, double *log_post,
                            uint32_t post_stride);

/* Device workspaces of finished calls are kept for the next call of similar shape (up to JTK_LC_POOL_GB per device,
 * default 32; 0 disables the pool): mapping tens of GB for a 2500-chunk batch costs seconds otherwise.  This returns
 * them to the driver. */
JTK_LC_API int jtk_lc_trim_cache(int device);

JTK_LC_API const char *jtk_lc_strerror(int status);
/* Thread-local text of the last failure on this thread (HIP error strings etc.); "" if none.  The pointer is valid until
 * this thread's next call into the library: copy the text before calling anything else. */
JTK_LC_API const char *jtk_lc_last_error(void);
JTK_LC_API int jtk_lc_version(void);
/* 1 if a gfx950 device `device` is present and the kernels for it are loaded. */
JTK_LC_API int jtk_lc_device_ok(int device);

/* Timing of the last jtk_lc_cluster_* call on this thread, measured with HIP events on the library's
 * own stream: total device milliseconds and the milliseconds + launch count of each kernel family
 * (index = enum jtk_kernel_id).  Used by bench.py for the live roofline number. */
enum jtk_kernel_id { JTK_K_PHMM = 0, JTK_K_POLISH = 1, JTK_K_FILTER = 2, JTK_K_MCMC = 3, JTK_K_COUNT = 4 };
typedef struct jtk_lc_timing {
    double total_ms;
    double h2d_ms, d2h_ms;
    double kernel_ms[JTK_K_COUNT];
    uint32_t kernel_launches[JTK_K_COUNT];
    uint32_t chain_lds_bytes[2]; /* LDS work area per chain workgroup of the two launch classes (0: class not used); class 0
                                  * stays <= 80 KiB (two workgroups per CU), class 1 <= 160 KiB */
} jtk_lc_timing_t;
JTK_LC_API int jtk_lc_last_timing(jtk_lc_timing_t *out);

#ifdef __cplusplus
}
#endif
#endif /* JTK_LC_H */
