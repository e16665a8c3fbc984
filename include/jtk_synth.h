/* jtk_synth.h -- host-only synthetic pile-up generator of bench.py and the tests, built as its own library
 * libjtk_synth.so (it is not linked into the product library libjtk_lc.so).  It stands in for the upstream JTK stages that produce the hot path's inputs
 * (chunk selection, minimap2/edlib encoding); the usage pattern follows the reference's dev harness
 * sandbox/src/bin/benchmark_clustering.rs:55-100 and gen_sim_genome.rs:24-28.  Not part of the hot path. */
#ifndef JTK_SYNTH_H
#define JTK_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct jtk_synth_cfg {
    uint64_t seed;          /* 20260101 + chunk_id in the BASELINE configs */
    uint32_t tmpl_len;      /* L */
    uint32_t n_haps;        /* number of distinct haplotypes / paralog copies */
    uint32_t reads_per_hap;
    uint32_t min_variants;  /* force at least this many variant events per non-reference haplotype */
    double divergence;      /* per-base variant rate of haplotype h>=1 vs haplotype 0 (1/3 per type) */
    double err_sub, err_ins, err_del; /* read error model */
    double tmpl_err;        /* residual per-base error of the chunk sequence vs haplotype 0 */
} jtk_synth_cfg_t;

/* One pile-up: n = n_haps*reads_per_hap reads in generation order (NOT yet in pileup_nodes order).
 * read_off/ops_off have n+1 entries; strand[n]; truth[n] = haplotype of each read. 0 or -1. */
__attribute__((visibility("default"))) int jtk_synth_pileup(const jtk_synth_cfg_t *cfg, uint8_t *tmpl, uint64_t tmpl_cap, uint64_t *tmpl_len_out,
                     uint8_t *reads, uint64_t reads_cap, uint64_t *read_off, uint8_t *ops, uint64_t ops_cap,
                     uint64_t *ops_off, uint8_t *strand, uint32_t *truth);

#ifdef __cplusplus
}
#endif
#endif
