// device_common.h -- shared declarations of the HIP kernels (gfx950 only) and the session that drives them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jtk_lc.h"
#include "jtk_math.h"

#define JTK_WAVE 64
#define JTK_MAX_RADIUS 30        // 2r+1 <= 61 lanes: 3 spare lanes make the lane ring unambiguous (phmm_kernel)
#define JTK_WIDE_MAX_RADIUS 255  // wider bands (CLR / None reads, long ONT chunks) take phmm_wide_kernel (LDS frames of 256 / 512 cells)
#define JTK_PAIR_MAX_RADIUS 14   // 2r+1 <= 29 cells + 3 spare lanes fit a 32-lane half: phmm_pair_kernel runs two reads per wave
#define JTK_SCALE_BLOCK 64       // one power-of-two exponent per 64 anti-diagonals (oracle/phmm.c)
#define JTK_SCRATCH_GUARD 16     // zero rows in front of a wave's forward stripe (phmm_kernel reads pairs of diagonals >= -13)
#define JTK_ACC_N 16             // accumulators per template row (see phmm_kernels.hip)
#define JTK_LOG_ZERO (-1.0e300)
#define JTK_LN2 0.6931471805599453094
#define JTK_POS_THR 0.00001      // pseudo_mcmc.rs:5
#define JTK_MASK_LENGTH 7        // pseudo_mcmc.rs:3
#define JTK_MAX_HOMOP_LENGTH 2   // pseudo_mcmc.rs:4
#define JTK_MAX_COPY 7           // one clustering() call sees copy_num < 8 (UPPER_COPY_NUM, mod.rs:85); larger chunks
                                 // go through clustering_recursive's split, which session.hip drives
#define JTK_TRACE_MAX_PICKS 64    // jtk_lc_session_trace: picks recorded per chunk (ROUND * max(copy_num, 2) <= 21 of them happen)
#define JTK_MAX_DIM (3 * JTK_MAX_COPY)  // ROUND * max(copy_num, 2) picked columns (pseudo_mcmc.rs:421,527,532)
#define JTK_MAX_PILEUP 1023      // reads per pile-up in the chain kernel: 10-bit read indices in its proposal records and hop
                                 // words (register tables to 255 reads, LDS tables beyond); the LDS work area (n x D
                                 // doubles + tables) has to fit 160 KiB as well: session.hip sizes it per launch
#define JTK_POLISH_MIN_GAIN 0.1
#define JTK_POLISH_MAX_ROUNDS 20

// One read of the resident batch.
struct ReadMeta {
    uint32_t chunk;      // index into ChunkMeta
    uint32_t read_len;   // n
    uint64_t ey_off;     // into d_ey: ey[j] = y[j-1] | ctx(j)<<2 for j = 1..n (ey[0] unused)
    uint64_t ops_off;    // into d_ops[cur] (capacity slot)
    uint32_t ops_cap;
    uint32_t strand;     // 1 = forward model
    uint64_t delta_off;  // into d_delta (u64 words)
    uint64_t table_off;  // the read's table, JTK_NUM_ROW doubles per position: == raw_off (finalize_kernel works in place)
    uint64_t raw_off;    // into d_raw (doubles): JTK_ACC_N * (tmpl_cap + 1) per read; row sums, then the table
    uint64_t row_off;    // into d_rawG (ints): tmpl_cap + 1 per read
};

// One chunk (pile-up) of the resident batch.
struct ChunkMeta {
    uint64_t chunk_id;
    uint32_t copy_num;
    uint32_t n_reads;
    uint32_t read_first;
    uint32_t tmpl_cap;   // capacity of each template buffer
    uint64_t tmpl_off;   // into d_tmpl[cur]
    uint64_t total_off;  // into d_total (doubles): JTK_NUM_ROW * (tmpl_cap + 1)
    uint32_t radius;     // band radius for this chunk: ceil(len0 * band_frac) / 2 (mod.rs:96,105)
    uint32_t edit_cap;   // capacity of this chunk's edit list
    uint64_t edit_off;   // into d_edits
    uint64_t feat_off;   // into d_feat (doubles): n_reads * JTK_MAX_DIM
    uint64_t cand_off;   // into d_cand (doubles): JTK_NUM_ROW * (tmpl_cap + 1) candidate scores
    double local_coverage;  // ClusteringConfig.local_coverage (mod.rs:108-112)
    uint32_t take_num;   // HMMPolishConfig take_num: only the first take_num reads vote in a polish round (0 = all)
    uint32_t pad0;
};

// Mutable per-chunk state (device resident).
struct ChunkState {
    uint32_t tmpl_len;   // current template length
    uint32_t buf;        // which of the two template / ops buffers is current
    uint32_t active;     // 1 while polishing has not converged
    uint32_t rounds;     // polish rounds executed
    int32_t status;      // jtk_status of this chunk
    uint32_t n_edits;    // edits selected in the current round
    uint32_t dim;        // D selected variant columns
    uint32_t k;          // cluster_num
    double score;
    uint64_t draws;      // RNG stream positions the clustering consumed
    uint64_t chain_cycles;  // shader-clock cycles the chunk's consumer wave spent in the chain kernel (jtk_lc_debug_chain_profile)
    uint32_t chain_events;  // proposals of its table-driven chains that could not be stepped over
    uint32_t chain_pad;
};

struct HmmDev {
    double a[9];      // mat_mat, mat_ins, mat_del, ins_mat, ins_ins, ins_del, del_mat, del_ins, del_del
    double eM[16];
    double eI[20];
};

struct Edit {
    uint32_t pos;
    uint32_t row;
};

// The three buffer sets (template codes, per-base ops and their lengths); ChunkState.buf selects the current one.  Set 0
// holds the batch as uploaded and is never written: a pass starts from it without a device-to-device reset copy; the first
// polish round that edits a chunk writes set 1, later rounds ping-pong between 1 and 2.  Passed to kernels by value.
struct DevBufs {
    uint8_t *tmpl[3];
    uint8_t *ops[3];
    uint32_t *ops_len[3];
};
#define JTK_NEXT_BUF(b) ((b) == 0u ? 1u : 3u - (b))

// ---- kernel launchers (definitions in the .hip files) ----
size_t phmm_lds_bytes(uint32_t max_tmpl, uint32_t max_read);
void launch_band_prep(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                      ChunkState *state, DevBufs bufs, uint64_t *delta, int only_active, uint32_t max_tmpl, uint32_t max_read);
// Work queues: `work_counter` is a device ticket counter that is never reset (no fill blit in front of a launch).  A wave
// takes tickets until one is past the launch's last item, so a launch advances the counter by exactly n_items + n_waves;
// `ticket_base` (host, owned by the session) is the counter's value when the launch starts and is advanced by the launcher.
// The forward scratch of the lane-ring pair-HMM kernels: stripes of (template + read + guard) x 1 KiB, one per RESIDENT wave.
// The set is shared by every session of a device (session.hip: StripePool): however many batches are in flight, no more
// waves than the device holds are ever inside a sweep, so a wave takes a free stripe when it starts and gives it back when it
// ends (owner[]: 0 = free).  Release / acquire at agent scope on the hand-over: a stripe may move between XCDs, whose L2s are
// not coherent with each other -- nothing depends on where a wave runs.
struct StripeSet {
    double *mem;
    uint64_t stride;  // doubles per stripe
    uint32_t *owner;
    uint32_t n;       // stripes (>= the waves the device can hold: nobody ever waits for long)
    uint32_t unfenced;  // measurement only (JTK_STRIPE_UNFENCED=1): hand over without the acquire / release fences
};
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t jtk_stripe_acquire(const StripeSet &ss) {
    uint32_t got = 0;
    if (threadIdx.x == 0) {
        // start anywhere (the cycle counter differs between waves and launches), probe linearly; holders never wait for anything,
        // so a full table only means a short wait
        uint32_t h = (uint32_t)((__builtin_readcyclecounter() * 0x9E3779B97F4A7C15ull) >> 40) % ss.n;
        for (uint32_t tries = 0;; tries++) {
            uint32_t expected = 0;
            if (__hip_atomic_compare_exchange_strong(&ss.owner[h], &expected, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT))
                break;
            if (++h == ss.n) h = 0;
            if ((tries & 255u) == 255u) __builtin_amdgcn_s_sleep(16);
        }
        got = h;
    }
    got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
    if (!ss.unfenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every lane: what the previous holder wrote is not read stale
    return got;
}
__device__ __forceinline__ void jtk_stripe_release(const StripeSet &ss, uint32_t stripe) {
    if (!ss.unfenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // every lane's stores to the stripe are out of this XCD's L2 first
    if (threadIdx.x == 0) __hip_atomic_store(&ss.owner[stripe], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif
void launch_phmm(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                 const ChunkState *state, DevBufs bufs, const uint8_t *ey, const uint64_t *delta,
                 const HmmDev *hmm2, StripeSet stripes, uint32_t n_waves,
                 uint32_t *work_counter, uint32_t *ticket_base, double *raw, int *rawG, double *lk, uint32_t max_tmpl,
                 uint32_t max_read, int only_active, uint32_t skip_le_radius = 0);
// phmm_pair.hip: the same sweep for band radius <= JTK_PAIR_MAX_RADIUS, two reads of a chunk per wave.  items[q] = index of
// the first read of the pair, bit 31 set when the item is a single read; phmm_kernel is then told to skip those chunks.
size_t phmm_pair_lds_bytes(uint32_t max_tmpl, uint32_t max_read);
void launch_phmm_pair(hipStream_t s, uint32_t n_items, const uint32_t *items, const ReadMeta *reads, const ChunkMeta *chunks,
                      const ChunkState *state, DevBufs bufs, const uint8_t *ey, const uint64_t *delta, const HmmDev *hmm2,
                      StripeSet stripes, uint32_t n_waves, uint32_t *work_counter, uint32_t *ticket_base,
                      double *raw, int *rawG, double *lk, uint32_t max_tmpl, uint32_t max_read, int only_active);
void launch_finalize(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                     const ChunkState *state, const HmmDev *hmm2, double *raw, const int *rawG, const double *lk,
                     uint32_t max_tmpl, int only_active, int converged_in = -1);  // in place: a read's table takes the place of
                                                                                   // its row sums; converged_in >= 0: only the
                                                                                   // chunks that converged in that polish round
// the column totals of a polish round straight from the row sums (finalize's arithmetic + sum_tables' sum, no table written)
void launch_sum_final(hipStream_t s, uint32_t n_chunks, const ReadMeta *reads, const ChunkMeta *chunks, const ChunkState *state,
                      const HmmDev *hmm2, const double *raw, const int *rawG, const double *lk, double *total, uint32_t max_tmpl);
// phmm_wide.hip: the reads of chunks whose band radius exceeds JTK_MAX_RADIUS (phmm_kernel skips them)
size_t phmm_wide_lds_bytes(uint32_t max_tmpl, uint32_t max_read, uint32_t max_radius);
uint64_t phmm_wide_scratch_doubles(uint32_t max_tmpl, uint32_t max_read, uint32_t max_radius);
void launch_phmm_wide(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                      const ChunkState *state, DevBufs bufs, const uint8_t *ey, const uint64_t *delta, const HmmDev *hmm2,
                      double *scratch, uint64_t scratch_stride, uint32_t n_waves, uint32_t *work_counter, uint32_t *ticket_base,
                      double *raw, int *rawG, double *lk, uint32_t max_tmpl, uint32_t max_read, int only_active,
                      uint32_t max_radius);
// expected transition / emission counts of every read of a batch (E-step of the model refit); 45 doubles per read
size_t phmm_counts_lds_bytes(uint32_t max_tmpl, uint32_t max_read, uint32_t max_radius);
uint64_t phmm_counts_scratch_doubles(uint32_t max_tmpl, uint32_t max_read, uint32_t max_radius);
void launch_phmm_counts(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                        const ChunkState *state, DevBufs bufs, const uint8_t *ey, const uint64_t *delta, const HmmDev *hmm2,
                        double *scratch, uint64_t scratch_stride, uint32_t n_waves, uint32_t *work_counter, uint32_t *ticket_base,
                        double *counts, double *lk, uint32_t max_tmpl, uint32_t max_read, uint32_t max_radius);
// io_kernels.hip: the raw reads / ops of a batch recoded on the device (flags: bit 1 = non-ACGT base, bit 2 = op code > 3), and
// the variable-length results packed for the copy back (lengths first, then consensus as ASCII + ops at prefix-summed offsets)
void launch_encode_reads(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const uint8_t *raw_bases,
                         const uint64_t *raw_base_off, const uint8_t *raw_ops, const uint64_t *raw_ops_off, uint8_t *ey,
                         uint8_t *ops, uint32_t *flags);
void launch_out_len(hipStream_t s, uint32_t n_reads, uint32_t n_chunks, const ReadMeta *reads, const ChunkState *state,
                    DevBufs bufs, uint32_t *ops_len_out, uint32_t *cons_len_out);
void launch_gather(hipStream_t s, uint32_t n_reads, uint32_t n_chunks, const ReadMeta *reads, const ChunkMeta *chunks,
                   const ChunkState *state, DevBufs bufs, const uint64_t *ops_out_off, const uint64_t *cons_off,
                   uint8_t *ops_out, uint8_t *cons_out);
// polish_kernels.hip
#define JTK_NACTIVE_SLOTS (JTK_POLISH_MAX_ROUNDS + 3)  // one "chunks still active" counter per polish round of a pass
void launch_reset_pass(hipStream_t s, uint32_t n_chunks, ChunkState *state, const ChunkState *state0, uint32_t *n_active,
                       uint32_t n_counters);
void launch_polish_round(hipStream_t s, uint32_t n_chunks, uint32_t n_reads, const ReadMeta *reads,
                         const ChunkMeta *chunks, ChunkState *state, DevBufs bufs, const uint8_t *ey,
                         const double *table, double *total, Edit *edits, uint32_t *new_len, uint32_t max_tmpl,
                         uint32_t ignore_edge, int final_pass, uint32_t *n_active_out, uint32_t *n_active_host);
                         // (`total` holds the round's column totals: launch_sum_final)
