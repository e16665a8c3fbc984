// mcmc_kernels.hip -- the read-clustering loop on the device: one workgroup of two wavefronts per chunk.
//
// Follows, statement by statement, haplotyper/src/local_clustering/pseudo_mcmc.rs
//   cluster_filtered_variants :213-274   mcmc_clustering :649-670   mcmc_with_filter :704-762
//   flip :764-783   get_lk :785-795   LKCount :797-845   get_used_columns :847-869
//   get_read_lk_gains :381-408   get_likelihood_gain :353-379   use_highest_gain :673-693
//   expected_gains :286-306   clustering tail :98-105   to_posterior_probability :342-347
// and haplotyper/src/misc.rs  kmeans :231-259, suggest_first :315-341, logsumexp :84-92,
// with the rand 0.8.5 / rand_xoshiro 0.6.0 sampling restated exactly as in oracle/rng.c.
//
// The chain is strictly sequential per chunk (one RNG stream threads through k-means, 20 restarts and every
// candidate k, local_clustering/mod.rs:97).  A lone wavefront on CDNA4 issues a dependent instruction every
// ~8 cycles and pays ~30 for a value that crosses from the vector to the scalar side, so the chain is bound by the
// length of the dependent chain of a step, not by bandwidth.  What buys speed without changing a bit of the result:
//  * PIPELINE: wave 1 ("producer") runs xoshiro256** lane-parallel (GF(2) jump-ahead) into an LDS ring and, for the
//    diploid chain, parses the proposal that would start at every stream position into a 32-bit record; wave 0
//    ("consumer") runs the algorithm and takes every random draw -- k-means initialisation, proposals, Bernoulli
//    tests -- from that ring, in stream order.  The stream never depends on the chain, so nothing is speculated.
//  * THE DIPLOID CHAIN (K == 2, n <= 127, D <= 8) keeps, per read, the exact likelihood of the state with that read
//    flipped (rebuilt lane-parallel after every move) and walks over the proposals that are certainly rejected and
//    leave no rounding residue; everything else is settled from the read's lane (mcmc_chain_k2).
//  * THE GENERIC CHAIN keeps LKCount[c][d] in registers (lane = column, K a template parameter), the counts as
//    integers (num_pos and 3*num_pos - 7*num_neg, which decides is_informative exactly), labels / sizes / >0 masks
//    as wave-uniform scalars, and commits or undoes a proposal with the reference's own arithmetic
//    ((tg - x) + x, not a restore); get_lk's left-to-right sum visits only the non-zero terms.
// Sums that the reference evaluates left to right are evaluated left to right here -- integer labels only
// match if every f64 rounding matches.
#include <cstdlib>
#include <type_traits>
#include <mutex>
#include <vector>

#include "device_common.h"

extern __shared__ __align__(16) unsigned char jtk_mcmc_smem[];  // the chain kernels' dynamic LDS (see lds_carve)

namespace {

// ---- LDS accessors for the producer/consumer hand-off.  The pointers reach us as generic pointers; casting
// them back to the LDS address space makes these ds_read/ds_write instead of waited flat accesses.
// volatile: re-read every time, in program order (LDS operations of one wave execute in order).
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
typedef __attribute__((address_space(3))) volatile uint64_t lds_vu64;
__device__ __forceinline__ uint32_t lds_ld32(const uint32_t *p) { return *(lds_vu32 *)p; }
__device__ __forceinline__ void lds_st32(uint32_t *p, uint32_t v) { *(lds_vu32 *)p = v; }
__device__ __forceinline__ uint64_t lds_ld64(const uint64_t *p) { return *(lds_vu64 *)p; }
__device__ __forceinline__ void lds_st64(uint64_t *p, uint64_t v) { *(lds_vu64 *)p = v; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}

// The ring of raw xoshiro256** outputs.  `wr` draws have been produced, the consumer has released `rd`.
#ifndef JTK_LIGHT_MAX_READS
#define JTK_LIGHT_MAX_READS 127u  // what mcmc_kernel_light takes: diploid pile-ups of <= 127 reads ...
#endif
#define JTK_LIGHT_MAX_DIM 2u      // ... with <= 2 variant columns (mcmc_chain_k2<1,..> / <2,..>: 168 registers)
// The producer's lanes generate SEG = 2^seg_log consecutive draws each per superblock (64 SEG draws); the ring holds two
// superblocks (RN draws + RN records).  seg_log is a launch parameter (round 5): 4 in the light kernel and the global-memory one
// (a 24 KiB ring: the jump is paid once per 1,024 draws), 3 in the general kernel -- a 12 KiB ring is what lets FOUR chain
// workgroups of a 4-copy pile-up share a CU's LDS (40.7 KB each), and the K-way chains' producer has slack for the extra jumps.
#ifndef JTK_SEG_LOG_LIGHT
#define JTK_SEG_LOG_LIGHT 4u
#endif
#define JTK_SEG_LOG_GENERAL 3u
#define RN_OF(seg_log) (128u << (seg_log))
#define JUMP_TAB_BYTES (128 * 4 * 32) // (what an LDS copy of the round-4 jump table took; the byte table lives in L2)
#define K2_STAT_SLOTS 24             // counters in LDS per chunk: 16 of the statistics build, [16] events (every build)
// stream position -> ring slot.  Inside a superblock, draw j of segment g sits at j * 64 + ((g + j) & 63): the
// producer's 64 lanes (one segment each) and the consumer's 64-draw windows (consecutive j) both hit distinct banks.
__device__ __forceinline__ uint32_t ring_slot(uint32_t pos, uint32_t seg_log) {
    const uint32_t half = 64u << seg_log;  // draws per superblock
    const uint32_t o = pos & (half - 1), g = o >> seg_log, j = o & ((1u << seg_log) - 1);
    return (pos & half) | (j * 64 + ((g + j) & 63));
}
struct RCtl {
    uint32_t rd, quit;  // written by the consumer (read together, 8-byte aligned)
    uint32_t wr, wp;    // written by the producer: draws produced / stream positions whose proposal record exists
    // Proposal records are parsed for ONE (format, K) at a time.  The consumer announces a new mode by writing parse_from
    // (the stream position from which it will read records), parse_n and then pmode = epoch << 16 | mode; the producer
    // re-parses from parse_from and acknowledges with wp_epoch = epoch (after resetting wp).
    uint32_t parse_n;     // reads in the pile-up
    uint32_t parse_from;
    uint32_t pmode;       // mode: 0 = no records, PM_K2 = the diploid chain's format, otherwise K of the general format
    uint32_t wp_epoch;
};
#define PM_K2 0x100u
// The consumer's view of the generator: a position in the stream of Xoshiro256StarStar::seed_from_u64(id * 3490)
// (local_clustering/mod.rs:97).  next_u64 == rand_xoshiro's next_u64, one stream position later.
struct Rng {
    uint32_t pos;      // next draw to take (absolute stream position)
    uint32_t wr_seen;  // producer progress last observed
    uint32_t wp_seen;  // record progress last observed
    uint32_t pmode;    // the parse mode last announced (epoch << 16 | mode)
    uint32_t win_base; // stream position of the draw held by lane 0 of `win`
    uint32_t seg_log;  // the ring's geometry (RN_OF(seg_log) draws)
    uint64_t win;      // per lane: the raw draw at win_base + lane (one LDS read serves 64 sequential draws)
#ifdef JTK_MCMC_STATS
    uint32_t waits;    // polls of the producer's counters that found nothing new
#endif
    RCtl *ctl;
    const uint64_t *ring;
    const uint32_t *rec;  // proposal records, one per stream position (see producer_parse)
};
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
__device__ __forceinline__ uint64_t splitmix64(uint64_t &x) {
    x += 0x9e3779b97f4a7c15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ void rng_wait(Rng &r, uint32_t upto) {  // until draws [.., upto) exist
    while ((int32_t)(r.wr_seen - upto) < 0) {
        r.wr_seen = uni(lds_ld32(&r.ctl->wr));
        if ((int32_t)(r.wr_seen - upto) < 0) __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void rng_wait_rec(Rng &r, uint32_t upto) {  // until records [.., upto) exist
    while ((int32_t)(r.wp_seen - upto) < 0) {
        r.wp_seen = uni(lds_ld32(&r.ctl->wp));
#ifdef JTK_MCMC_STATS
        if ((int32_t)(r.wp_seen - upto) < 0) r.waits++;
#endif
        if ((int32_t)(r.wp_seen - upto) < 0) __builtin_amdgcn_s_sleep(1);
    }
}
// The records from stream position r.pos on are wanted in `mode` (see RCtl); returns once the producer has switched.
__device__ __forceinline__ void rng_set_parse_mode(Rng &r, uint32_t mode, uint32_t lane) {
    if ((r.pmode & 0xffffu) == mode) return;  // the producer parses every position: nothing to re-synchronise
    const uint32_t word = (((r.pmode >> 16) + 1u) << 16) | mode;
    r.pmode = word;
    if (lane == 0) {
        lds_st32(&r.ctl->rd, r.pos);
        lds_st32(&r.ctl->parse_from, r.pos);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lds_st32(&r.ctl->pmode, word);
    while (uni(lds_ld32(&r.ctl->wp_epoch)) != (word >> 16)) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    r.wp_seen = r.pos;  // progress of the old mode says nothing about the new one
}
__device__ __forceinline__ void rng_release(Rng &r, uint32_t lane) {  // draws before r.pos may be overwritten
    if (lane == 0) lds_st32(&r.ctl->rd, r.pos);
}
__device__ __forceinline__ void rng_refill(Rng &r) {  // the register window: 64 draws from r.pos on, one per lane
    r.win_base = r.pos;
    lds_st32(&r.ctl->rd, r.pos);  // every lane stores the same value: draws before r.pos may be overwritten
    rng_wait(r, r.pos + 64);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    r.win = lds_ld64(&r.ring[ring_slot(r.pos + (threadIdx.x & 63u), r.seg_log)]);
}
__device__ __forceinline__ uint64_t next_u64(Rng &r) {
    if ((uint32_t)(r.pos - r.win_base) >= 64u) rng_refill(r);
    const uint32_t off = r.pos - r.win_base;
    const uint64_t v = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(r.win >> 32), (int)off) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)r.win, (int)off);
    r.pos++;
    return v;
}
__device__ __forceinline__ uint32_t next_u32(Rng &r) { return (uint32_t)(next_u64(r) >> 32); }
__device__ __forceinline__ uint64_t gen_range_usize(Rng &r, uint64_t n) {
    const uint64_t zone = (n << __clzll((long long)n)) - 1;
    for (;;) {
        const uint64_t v = next_u64(r);
        const uint64_t hi = __umul64hi(v, n), lo = v * n;
        if (lo <= zone) return hi;
    }
}
__device__ __forceinline__ uint32_t gen_range_u32(Rng &r, uint32_t n) {
    const uint32_t zone = (n << __clz((int)n)) - 1;
    for (;;) {
        const uint32_t v = next_u32(r);
        const uint64_t m = (uint64_t)v * n;
        if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
    }
}
__device__ __forceinline__ uint64_t gen_index(Rng &r, uint64_t ub) {
    return ub <= 0xffffffffULL ? gen_range_u32(r, (uint32_t)ub) : gen_range_usize(r, ub);
}
__device__ __forceinline__ bool gen_bool(Rng &r, double p) {
    if (p == 1.0) return true;
    const double scaled = p * 18446744073709551616.0;
    const uint64_t p_int = !(scaled > 0.0) ? 0ull : __double2ull_rz(scaled);
    return next_u64(r) < p_int;
}
__device__ __forceinline__ uint32_t choose_other(Rng &r, uint32_t k, uint32_t old) {
    uint32_t result = 0xffffffffu, consumed = 0;
    for (uint32_t c = 0; c < k; c++) {
        if (c == old) continue;
        consumed++;
        if (gen_index(r, consumed) == 0) result = c;
    }
    return result;
}

// Only wave 0 runs the non-chain phases, so LDS hand-offs between its lanes need a wave-level fence, not a
// workgroup barrier (the producer wave is parked at a real barrier meanwhile).
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ double unif64(double v) { return jtk_bits_f64(uni64(jtk_f64_bits(v))); }
// a wave-uniform condition as a scalar: branches on it are s_cbranch, not exec-mask regions
__device__ __forceinline__ bool ubool(bool c) { return __ballot(c) != 0ull; }  // c is the same in every lane

// LDS work area of one chunk
struct Elem {  // one (read, column) cell as the chain needs it (derived from the value on the fly: LDS holds only x)
    double x;  // the likelihood gain
    int dp;    // 1 if x >  POS_THR (counts towards num_pos)
    int pw;    // 3*[x > POS_THR] - 7*[x < -POS_THR]: increment of 3*num_pos - 7*num_neg
};
__device__ __forceinline__ Elem elem_of(double x) {
    Elem el;
    el.x = x;
    el.dp = JTK_POS_THR < x ? 1 : 0;
    el.pw = 3 * el.dp - 7 * (x < -JTK_POS_THR ? 1 : 0);
    return el;
}
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// The work area is carved out of the workgroup's dynamic LDS from four numbers.  Out-of-line functions get THOSE (by value)
// and carve again: two dozen 64-bit pointers that live across a call -- or travel through memory as a struct -- cost the
// caller its registers and the callee flat loads (and hipcc 7.2 fails with "Subtarget requires even aligned vector
// registers" when a kernel body has to spill a 64-bit value across a call).
struct LdsShape {
    uint32_t n, d, k, seg_log;  // capacity in reads / columns / clusters; the ring's geometry
    uint64_t gws;            // 0, or the address of a global-memory workspace (mcmc_kernel_huge): whatever of the arrays sized
                             // by n / d / k does not fit the JTK_HUGE_LDS bytes of LDS behind the ring goes there, in carve order
};
#define JTK_HUGE_LDS (128u * 1024u)  // dynamic LDS of mcmc_kernel_huge behind the ring and its control block
struct Lds {
    RCtl *ctl;
    uint64_t *ring;      // RN raw draws
    uint32_t *rec;       // RN proposal records of the diploid chain
    ulonglong2 *jump;    // the producer's jump table (JUMP_TAB_BYTES)
    unsigned long long *k2_stats;  // 16 debug counters (JTK_MCMC_STATS builds only) + [16]: events of the table-driven chains
    double *data;        // n x D
    double *size_to_lk;  // n + 1
    double *lfact;       // n + 1
    double *val;         // K x D staging of the per-(cluster, column) terms
    double *centers;     // K x D
    double *fbuf;        // n (dists / weights / per-read gains)   -- fbuf and cum alias the head of stab (lds_carve)
    double *cum;         // n
    uint8_t *assign;     // n   current labels
    uint8_t *argmax;     // n   best labels seen in this chain
    uint8_t *best;       // n   best over restarts for this k
    uint8_t *accepted;   // n   labels of the accepted k
    uint8_t *used;       // D
    uint8_t *prev_used;  // D
    uint8_t *tmp_asn;    // n
    uint8_t *tmp_used;   // D
    // tables of the table-driven chain (mcmc_chain_tab)
    double *stab;               // lds_k x npad: s[c][i] = sum over the columns cluster c is paid for of x[i][d]
    uint32_t *nz;               // npad: bit d set iff x[i][d] != 0.0
    struct SzEnt *sz;           // K per-cluster terms: size deltas and the columns where a move involving c is not certified
    u32x4_t *st;                // D x K: (total_gain, num_pos, 3 num_pos - 7 num_neg) of (column, cluster), 16 bytes each
    u32x4_t *col;               // D: (pos_in_use, informative clusters, total pos, -) of the column
    uint32_t npad;              // row stride of stab
    struct LdsShape shape;      // what the carve was made from
};
struct SzEnt {     // 32 bytes, read as two 16-byte vectors
    double rem;    // size_to_lk[size - 1] - size_to_lk[size]: what the size terms gain when a read leaves the cluster
    double add;    // size_to_lk[size + 1] - size_to_lk[size]
    uint32_t nr;   // columns where this cluster's `total_gain > 0` could flip under a single move
    uint32_t um;   // columns this cluster is paid for (used and total_gain > 0): what s[c][.] sums over
    uint32_t pad[2];
};

// The carve (host twin: mcmc_lds_core).  `base` passes through an empty asm so that two carves are not merged across a call.
template <bool HUGE = false>
__device__ __forceinline__ Lds lds_carve(LdsShape sh_in) {
    LdsShape sh;
    sh.n = uni(sh_in.n);
    sh.d = uni(sh_in.d);
    sh.k = uni(sh_in.k);
    sh.seg_log = uni(sh_in.seg_log);
    sh.gws = uni64(sh_in.gws);
    uint32_t base = 0;
    asm volatile("" : "+s"(base));
    unsigned char *p = jtk_mcmc_smem + base;
    // mcmc_kernel_huge only: an array that does not fit what is left of JTK_HUGE_LDS lives in the chunk's global workspace.
    // There every pointer is made from an integer that went through an empty asm: the optimizer must not try to prove an address
    // space for a pointer that is LDS on one path and global on the other (hipcc 7.2 crashes in simplifycfg when it does).
    uint64_t pl = 0, gl = sh.gws;
    size_t lds_left = ~(size_t)0;
    if (HUGE) {
        pl = (uint64_t)(uintptr_t)p;
        asm volatile("" : "+s"(pl));
    }
    auto take = [&](size_t bytes) -> unsigned char * {
        bytes = (bytes + 15) & ~(size_t)15;
        if (HUGE) {
            const bool in_lds = bytes <= lds_left;
            uint64_t q = in_lds ? pl : gl;
            pl += in_lds ? bytes : 0;
            gl += in_lds ? 0 : bytes;
            lds_left -= in_lds ? bytes : 0;
            asm volatile("" : "+s"(q));
            return reinterpret_cast<unsigned char *>((uintptr_t)q);
        }
        unsigned char *q = p;
        p += bytes;
        return q;
    };
    const uint32_t lds_n = sh.n, lds_d = sh.d, lds_k = sh.k;
    Lds m;
    m.ctl = (RCtl *)take(sizeof(RCtl));
    m.ring = (uint64_t *)take(sizeof(uint64_t) * RN_OF(sh.seg_log));
    m.rec = (uint32_t *)take(sizeof(uint32_t) * RN_OF(sh.seg_log));
    m.jump = nullptr;
    m.k2_stats = (unsigned long long *)take(K2_STAT_SLOTS * 8);
    if (HUGE) lds_left = JTK_HUGE_LDS;  // from here on an array that does not fit goes to the workspace (host twin: mcmc_ws_bytes)
    m.data = (double *)take((size_t)lds_n * lds_d * 8);
    m.size_to_lk = (double *)take((size_t)(lds_n + 1) * 8);
    m.lfact = (double *)take((size_t)(lds_n + 1) * 8);
    m.val = (double *)take((size_t)JTK_MAX_COPY * lds_d * 8);
    m.centers = (double *)take((size_t)JTK_MAX_COPY * lds_d * 8);
    m.assign = (uint8_t *)take(lds_n);
    m.argmax = (uint8_t *)take(lds_n);
    m.best = (uint8_t *)take(lds_n);
    m.accepted = (uint8_t *)take(lds_n);
    m.tmp_asn = (uint8_t *)take(lds_n);
    m.used = (uint8_t *)take(lds_d);
    m.prev_used = (uint8_t *)take(lds_d);
    m.tmp_used = (uint8_t *)take(lds_d);
    m.npad = (lds_n + 63u) & ~63u;
    m.stab = (double *)take((size_t)lds_k * m.npad * 8);
    // The k-means scratch (and, in its place, the diploid chain's 16-byte entries) shares the first 16 n bytes of stab: stab is
    // rebuilt by the first publish() of every K-way chain (umask starts as "never built") and nothing reads it between chains,
    // k-means and get_read_lk_gains run only between them.  2.5 KB per chunk at 160 reads -- what a 4-copy pile-up's work area
    // (55.7 KB) was above a third of a CU's LDS: three chain workgroups per CU instead of two (cfg 4).
    m.fbuf = m.stab;
    m.cum = m.stab + lds_n;
    m.nz = (uint32_t *)take((size_t)m.npad * 4);
    m.sz = (SzEnt *)take((size_t)lds_k * sizeof(SzEnt));
    m.st = (u32x4_t *)take((size_t)lds_d * lds_k * 16);
    m.col = (u32x4_t *)take((size_t)lds_d * 16);
    m.shape = sh;
    return m;
}

// slice.choose_weighted over weights w[0..n) in LDS; cum is scratch. Returns -1 on WeightedError.
__device__ __forceinline__ int choose_weighted(Rng &r, const double *w, uint32_t n, double *cum, uint32_t lane) {
    double total = w[0];
    if (!(total >= 0.0)) return -1;
    bool bad = false;
    for (uint32_t i = 1; i < n; i++) {
        const double wi = w[i];
        if (!(wi >= 0.0)) bad = true;
        if (lane == 0) cum[i - 1] = total;
        total += wi;
    }
    if (bad || total == 0.0) return -1;
    double scale = total;
    const double max_rand = 1.0 - 0x1p-52;
    while (scale * max_rand + 0.0 >= total) scale = jtk_bits_f64(jtk_f64_bits(scale) - 1);
    const double v12 = jtk_bits_f64((next_u64(r) >> 12) | 0x3ff0000000000000ULL);
    const double chosen = (v12 - 1.0) * scale + 0.0;
    wsync();
    // partition point of `cum[i] <= chosen` (cum is non-decreasing): count the entries <= chosen
    uint32_t cnt = 0;
    for (uint32_t i = lane; i + 1 < n; i += 64) cnt += cum[i] <= chosen ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    wsync();
    return (int)cnt;
}

__device__ __forceinline__ double dist_row(const double *a, const double *b, uint32_t D) {
    double s = 0.0;
    for (uint32_t d = 0; d < D; d++) {
        const double t = a[d] - b[d];
        s += t * t;
    }
    return s;
}

// misc.rs:261-276 with centres given as K rows of D doubles in LDS (first minimum wins)
__device__ __forceinline__ void update_assignments(const Lds &m, uint32_t n, uint32_t D, uint32_t k, const double *centers,
                                   uint8_t *assign, uint32_t lane) {
    for (uint32_t i = lane; i < n; i += 64) {
        uint32_t best = 0;
        double bd = dist_row(m.data + i * D, centers, D);
        for (uint32_t c = 1; c < k; c++) {
            const double d = dist_row(m.data + i * D, centers + c * D, D);
            if (d < bd) {
                bd = d;
                best = c;
            }
        }
        assign[i] = (uint8_t)best;
    }
    wsync();
}

// misc.rs:298-307: sum over reads, in read order, of dist(read, its centre)
__device__ __forceinline__ double get_dist(const Lds &m, uint32_t n, uint32_t D, const uint8_t *assign, uint32_t lane) {
    for (uint32_t i = lane; i < n; i += 64) m.fbuf[i] = dist_row(m.data + i * D, m.centers + assign[i] * D, D);
    wsync();
    double s = 0.0;
    for (uint32_t i = 0; i < n; i++) s += m.fbuf[i];
    wsync();
    return s;
}

// misc.rs:229-259; returns false where the reference would panic
__device__ __forceinline__ bool kmeans(const Lds &m, uint32_t n, uint32_t D, uint32_t k, Rng &rng, uint32_t lane) {
    const double UPDATE_THR = 0.00000001;
    if (gen_bool(rng, 0.5)) {
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t c = (uint32_t)gen_range_usize(rng, k);
            if (lane == 0) m.assign[i] = (uint8_t)c;
        }
        wsync();
    } else {
        // suggest_first (misc.rs:315-341): centre rows are borrowed data rows; keep their indices in cum's tail
        uint32_t centre_idx[JTK_MAX_COPY];
        centre_idx[0] = (uint32_t)gen_index(rng, n);
        uint32_t nc = 1;
        for (uint32_t it = 0; it + 1 < k; it++) {
            for (uint32_t i = lane; i < n; i += 64) {
                double mn = dist_row(m.data + i * D, m.data + centre_idx[0] * D, D);
                for (uint32_t c = 1; c < nc; c++) {
                    const double d = dist_row(m.data + i * D, m.data + centre_idx[c] * D, D);
                    if (d < mn) mn = d;
                }
                m.fbuf[i] = mn;
            }
            wsync();
            const int idx = choose_weighted(rng, m.fbuf, n, m.cum, lane);
            if (idx < 0) return false;
            centre_idx[nc++] = (uint32_t)idx;
        }
        for (uint32_t c = 0; c < k; c++)
            for (uint32_t d = lane; d < D; d += 64) m.centers[c * D + d] = m.data[centre_idx[c] * D + d];
        wsync();
        update_assignments(m, n, D, k, m.centers, m.assign, lane);
    }
    // Lloyd iterations; `dist` is first evaluated against all-zero centres
    for (uint32_t e = lane; e < k * D; e += 64) m.centers[e] = 0.0;
    wsync();
    double dist = get_dist(m, n, D, m.assign, lane);
    for (;;) {
        // update_centers (misc.rs:277-297): per (cluster, column) slot, sum in read order
        for (uint32_t e = lane; e < k * D; e += 64) {
            const uint32_t c = e / D, d = e % D;
            double s = 0.0;
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < n; i++)
                if (m.assign[i] == c) {
                    s += m.data[i * D + d];
                    cnt++;
                }
            m.centers[e] = cnt > 0 ? s / (double)cnt : s;
        }
        wsync();
        update_assignments(m, n, D, k, m.centers, m.assign, lane);
        const double nd = get_dist(m, n, D, m.assign, lane);
        if (!(nd < dist + UPDATE_THR)) return false;  // assert!(new_dist < dist + UPDATE_THR)
        if (dist - nd < UPDATE_THR) break;
        dist = nd;
    }
    return true;
}

// Per-lane LKCount of one column for K clusters.
template <int K>
struct Counts {
    double tg[K];
    int np[K], nn[K];
};

template <int K>
__device__ __forceinline__ void lk_add(Counts<K> &q, uint32_t c, double x) {
#pragma unroll
    for (int cc = 0; cc < K; cc++)
        if ((uint32_t)cc == c) {  // c is wave-uniform: a scalar branch, static register index
            q.tg[cc] += x;
            if (JTK_POS_THR < x)
                q.np[cc]++;
            else if (x < -JTK_POS_THR)
                q.nn[cc]++;
        }
}
template <int K>
__device__ __forceinline__ void lk_sub(Counts<K> &q, uint32_t c, double x) {
#pragma unroll
    for (int cc = 0; cc < K; cc++)
        if ((uint32_t)cc == c) {
            q.tg[cc] -= x;
            if (JTK_POS_THR < x)
                q.np[cc]--;
            else if (x < -JTK_POS_THR)
                q.nn[cc]--;
        }
}

// get_used_columns (:847-869) for this lane's column.
// LKCount::is_informative (:818-822) is `0 < total_gain && 0.70 < num_pos / (num_pos + num_neg + 1e-7)`; for
// integer counts the f64 quotient test is exactly `3*num_pos > 7*num_neg` (no count pair comes within 1e-10 of
// the threshold; tests/test_host_and_abi.py checks every pair up to 2000 against the f64 expression).
template <int K>
__device__ __forceinline__ bool column_used(const Counts<K> &q) {
    bool any = false;
    int in_use = 0, in_neg = 0;
#pragma unroll
    for (int c = 0; c < K; c++) {
        const bool pos = 0.0 < q.tg[c];
        any |= pos && 3 * q.np[c] > 7 * q.nn[c];
        in_use += pos ? q.np[c] : 0;
        in_neg += pos ? 0 : q.np[c];
    }
    return any && 2 * in_neg < in_use;
}

template <int K>
__device__ __forceinline__ void fill_counts(const Lds &m, uint32_t n, uint32_t D, const uint8_t *assign,
                                            Counts<K> &q, int *clusters, uint32_t lane) {
#pragma unroll
    for (int c = 0; c < K; c++) {
        q.tg[c] = 0.0;
        q.np[c] = 0;
        q.nn[c] = 0;
        clusters[c] = 0;
    }
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = uni(assign[i]);
        const double x = lane < D ? m.data[i * D + lane] : 0.0;
        lk_add<K>(q, c, x);
#pragma unroll
        for (int cc = 0; cc < K; cc++)
            if ((uint32_t)cc == c) clusters[cc]++;
    }
}

__device__ __forceinline__ double readlane_f64(double v, uint32_t l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), (int)l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), (int)l);
    return __hiloint2double(hi, lo);
}

// Per-read values spread over lanes: element i lives in lane i & 63 of register i >> 6 (n <= 255).
// SMALL (n <= 63): everything sits in register 0 and the lookups are branch-free.
struct LaneTab {
    double v[4];
};
template <bool SMALL>
__device__ __forceinline__ double tab_get(const LaneTab &t, uint32_t i) {
    if (SMALL) return readlane_f64(t.v[0], i);
    const uint32_t l = i & 63;
    switch (i >> 6) {
        case 0: return readlane_f64(t.v[0], l);
        case 1: return readlane_f64(t.v[1], l);
        case 2: return readlane_f64(t.v[2], l);
        default: return readlane_f64(t.v[3], l);
    }
}
struct LaneLabels {
    int v[4];
};
template <bool SMALL>
__device__ __forceinline__ uint32_t lab_get(const LaneLabels &a, uint32_t i) {
    if (SMALL) return (uint32_t)__builtin_amdgcn_readlane(a.v[0], (int)i);
    const int l = (int)(i & 63);
    switch (i >> 6) {
        case 0: return (uint32_t)__builtin_amdgcn_readlane(a.v[0], l);
        case 1: return (uint32_t)__builtin_amdgcn_readlane(a.v[1], l);
        case 2: return (uint32_t)__builtin_amdgcn_readlane(a.v[2], l);
        default: return (uint32_t)__builtin_amdgcn_readlane(a.v[3], l);
    }
}
template <bool SMALL>
__device__ __forceinline__ void lab_set(LaneLabels &a, uint32_t i, uint32_t val, uint32_t lane) {
    const bool mine = lane == (i & 63);
    if (SMALL) {
        a.v[0] = mine ? (int)val : a.v[0];
        return;
    }
    switch (i >> 6) {
        case 0: a.v[0] = mine ? (int)val : a.v[0]; break;
        case 1: a.v[1] = mine ? (int)val : a.v[1]; break;
        case 2: a.v[2] = mine ? (int)val : a.v[2]; break;
        default: a.v[3] = mine ? (int)val : a.v[3]; break;
    }
}

// Position (0-based among the K-1 candidates) that `(0..K).filter(|c| c != old).choose(rng)` selects
// (pseudo_mcmc.rs:732): the i-th yielded candidate replaces the pick iff gen_index(i) == 0, whatever `old` is.
__device__ __forceinline__ uint32_t choose_pos(Rng &r, uint32_t k) {
    uint32_t pos = 0;
    for (uint32_t i = 1; i < k; i++)
        if (gen_index(r, i) == 0) pos = i - 1;
    return pos;
}

// The producer wave: Xoshiro256StarStar::seed_from_u64(seed), free running into the ring.
//
// xoshiro's state update is linear over GF(2), so the stream can be cut into segments that are generated side by
// side: lane l of the producer owns segment l of the current superblock (SEG consecutive draws) and runs the plain
// generator on its own copy of the state with ordinary 64-bit vector arithmetic -- 64 draws per ~20 instructions
// instead of one draw per ~11 scalar instructions.  After a superblock every lane stands at the start of the NEXT
// lane's segment and has to skip the other 63 segments: multiplication of the 256-bit state by the constant matrix
// M^(63*SEG), done as 128 two-bit look-ups in a 16 KiB table (g_jump_tab, computed once on the host from the step
// function itself, staged in LDS) XOR-ed together.  The sequence of draws is exactly that of the sequential generator.
// [byte of the state][value of that byte] -> 256-bit image under M^(63*SEG): 256 KiB in device memory, read by every producer
// wave of the machine (L2 resident).  One jump is 32 look-ups of 32 bytes XOR-ed together; up to round 4 the digits had two bits
// (128 look-ups in a 16 KiB table): 12 cycles per draw, as much as parsing the proposals -- now 4.
__device__ ulonglong2 g_jump_tab[2][32 * 256 * 2];  // [seg_log - 3]: M^(63 * 8), M^(63 * 16)

struct Xo {
    uint64_t s0, s1, s2, s3;
};
__device__ __forceinline__ void xo_step(Xo &x) {
    const uint64_t t = x.s1 << 17;
    x.s2 ^= x.s0;
    x.s3 ^= x.s1;
    x.s1 ^= x.s2;
    x.s0 ^= x.s3;
    x.s2 ^= t;
    x.s3 = rotl64(x.s3, 45);
}
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
// The loop is compact on purpose: fully unrolled it is kilobytes of straight-line code executed once per superblock, and
// this kernel is large.  Eight look-ups (16 loads) are in flight at a time.
__device__ __forceinline__ void xo_jump(Xo &x, uint32_t seg_log) {
    const u64x2 *tab = reinterpret_cast<const u64x2 *>(g_jump_tab[seg_log - 3u]);
    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
        const uint64_t wq = q == 0 ? x.s0 : (q == 1 ? x.s1 : (q == 2 ? x.s2 : x.s3));
        const u64x2 *row = tab + (size_t)(q * 8) * 256 * 2;
        u64x2 lo[8], hi[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t v = (uint32_t)(wq >> (8 * u)) & 255u;
            const u64x2 *e = row + ((size_t)u * 256 + v) * 2;
            lo[u] = e[0];
            hi[u] = e[1];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            a0 ^= lo[u].x;
            a1 ^= lo[u].y;
            a2 ^= hi[u].x;
            a3 ^= hi[u].y;
        }
    }
    x.s0 = a0;
    x.s1 = a1;
    x.s2 = a2;
    x.s3 = a3;
}
// Proposal records.  For the diploid chain a proposal is: gen_range(0..n) takes the first draw at or after its start
// whose widening multiply is accepted, gen_index(1) (the single candidate of K == 2, pseudo_mcmc.rs:732) then takes
// draws until one has a clear top bit, and the next draw is the one a Bernoulli test would compare.  None of this
// depends on the chain, so the producer parses the proposal that WOULD start at every stream position q:
//   rec[q] = idx | len << 7 | (top 19 bits of the Bernoulli draw) << 13      (idx < 128; len = draws used incl. that draw)
// rec == 0: not parsed (needs more than the 16..63 draws of look-ahead; the consumer then steps with scalar draws).
// 64 positions are parsed at once -- acceptance masks by ballot, "next accepted draw at or after p" by s_ff1 -- and
// the first PKEEP are kept, so every kept start had at least 64 - PKEEP draws of look-ahead (a proposal needs more
// with probability 2^-14).  R rounds are written stage by stage so that their instruction streams interleave:
// a lone wave pays ~8 cycles for a dependent instruction and ~4 for an independent one.
#define PKEEP 48
template <int R>
__device__ __forceinline__ void producer_parse(const uint64_t *ring, uint32_t *rec, uint32_t base, uint32_t n, uint32_t lane,
                                               uint32_t seg_log) {
    const uint64_t zone = ((uint64_t)n << __clzll((long long)n)) - 1;
    uint64_t draw[R];
    uint32_t hi[R], pi[R], pv[R], idx[R], vhi[R];
    bool ok[R];
#pragma unroll
    for (int r = 0; r < R; r++) draw[r] = lds_ld64(&ring[ring_slot(base + r * PKEEP + lane, seg_log)]);
#pragma unroll
    for (int r = 0; r < R; r++) {
        hi[r] = (uint32_t)__umul64hi(draw[r], (uint64_t)n);
        const unsigned long long okm = __ballot(draw[r] * (uint64_t)n <= zone);  // gen_range(0..n) accepts this draw
        const unsigned long long topm = __ballot((int64_t)draw[r] >= 0);          // gen_index(1) accepts this draw
        const unsigned long long m1 = okm >> lane;
        pi[r] = lane + (uint32_t)__builtin_ctzll(m1 | (1ull << 63));
        const unsigned long long m2 = pi[r] < 63 ? topm >> (pi[r] + 1) : 0ull;
        pv[r] = pi[r] + 1 + (uint32_t)__builtin_ctzll(m2 | (1ull << 63)) + 1;  // the Bernoulli draw
        ok[r] = m1 != 0 && m2 != 0 && pv[r] < 64;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        idx[r] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((pi[r] & 63) << 2), (int)hi[r]);
        vhi[r] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((pv[r] & 63) << 2), (int)(uint32_t)(draw[r] >> 32));
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t v = ok[r] ? (idx[r] | ((pv[r] + 1 - lane) << 7) | (vhi[r] & 0xffffe000u)) : 0u;
        if (lane < PKEEP) lds_st32(&rec[(base + r * PKEEP + lane) & (RN_OF(seg_log) - 1)], v);
    }
}
// Records of the general chain (any K).  A proposal is gen_range(0..n) -- the first draw at or after its start whose
// widening product passes the zone test -- then gen_index(i) for i = 1..K-1 on the upper halves of the following draws,
// each with its own zone test (IteratorRandom::choose over the K-1 other clusters, pseudo_mcmc.rs:732: the pick is the
// last i whose index came out 0), and the next draw is the one a Bernoulli test would compare:
//   rec[q] = idx (10 bits) | pick << 10 (3) | len << 13 (6: draws used incl. the Bernoulli draw) | top 13 bits of that draw << 19
// rec == 0: not parsed (needs more look-ahead than the window gives).  `keep` positions are kept per round, so every kept
// start had 64 - keep draws of look-ahead.
__device__ __forceinline__ void producer_parse_gen(const uint64_t *ring, uint32_t *rec, uint32_t base, uint32_t n, uint32_t K,
                                                   uint32_t keep, uint32_t lane, uint32_t seg_log) {
    const uint64_t zone = ((uint64_t)n << __clzll((long long)n)) - 1;
    const uint64_t draw = lds_ld64(&ring[ring_slot(base + lane, seg_log)]);
    const uint32_t v32 = (uint32_t)(draw >> 32);
    const uint32_t hi = (uint32_t)__umul64hi(draw, (uint64_t)n);
    const unsigned long long ok0 = __ballot(draw * (uint64_t)n <= zone);
    const unsigned long long m0 = ok0 >> lane;
    bool good = m0 != 0ull;
    uint32_t p = lane + (uint32_t)__builtin_ctzll(m0 | (1ull << 63));  // window offset of the gen_range draw
    const uint32_t idx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((p & 63u) << 2), (int)hi);
    uint32_t pick = 0;
    for (uint32_t i = 1; i < K; i++) {
        const uint32_t zi = (i << __builtin_clz(i)) - 1u;
        const uint64_t mi = (uint64_t)v32 * i;
        const unsigned long long okm = __ballot((uint32_t)mi <= zi);
        const unsigned long long zm = __ballot((uint32_t)(mi >> 32) == 0u);
        const unsigned long long mm = (good && p < 63u) ? okm >> (p + 1u) : 0ull;
        good = good && mm != 0ull;
        p = (p + 1u + (uint32_t)__builtin_ctzll(mm | (1ull << 63))) & 127u;
        if (good && ((zm >> (p & 63u)) & 1ull)) pick = i - 1u;
    }
    const uint32_t pv = p + 1u;  // the Bernoulli draw
    good = good && pv < 64u;
    const uint32_t vhi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((pv & 63u) << 2), (int)v32);
    const uint32_t v = good ? (idx | (pick << 10) | ((pv + 1u - lane) << 13) | (vhi & 0xfff80000u)) : 0u;
    if (lane < keep) lds_st32(&rec[(base + lane) & (RN_OF(seg_log) - 1)], v);
}
__device__ __forceinline__ void producer_main(RCtl *ctl, uint64_t *ring, uint32_t *rec, uint32_t seg_log, uint64_t seed,
                                              const uint64_t *resume, uint32_t lane) {
    const uint32_t SEG = 1u << seg_log, SBLK = 64u << seg_log, RN = RN_OF(seg_log);
    uint64_t z = seed;
    Xo x;
    if (resume) {  // a later clustering() call of the same chunk continues the stream (clustering_recursive, mod.rs:158)
        x.s0 = resume[0];
        x.s1 = resume[1];
        x.s2 = resume[2];
        x.s3 = resume[3];
    } else {
        x.s0 = splitmix64(z);
        x.s1 = splitmix64(z);
        x.s2 = splitmix64(z);
        x.s3 = splitmix64(z);
    }
    for (uint32_t j = 0; j < lane * SEG; j++) xo_step(x);  // lane l starts at stream position l * SEG
    uint32_t parse_n = 0, pmode = 0;
    uint32_t wr = 0, wp = 0;
#ifdef JTK_MCMC_STATS
    uint32_t st_sleeps = 0;
    unsigned long long st_gen = 0, st_parse = 0, st_jump = 0;
#endif
    for (;;) {
        const uint64_t c = uni64(lds_ld64((const uint64_t *)&ctl->rd));  // rd, quit
        if ((uint32_t)(c >> 32)) {
#ifdef JTK_MCMC_STATS
            if (lane == 0) printf("K2PROD wr %u sleeps %u cyc_gen %llu cyc_parse %llu cyc_jump %llu\n", wr, st_sleeps, st_gen, st_parse, st_jump);
#endif
            return;
        }
        {   // a new parse mode: records are re-parsed from the position the consumer names
            const uint32_t pm = uni(lds_ld32(&ctl->pmode));
            if (pm != pmode) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                pmode = pm;
                parse_n = uni(lds_ld32(&ctl->parse_n));
                wp = uni(lds_ld32(&ctl->parse_from));
                if (lane == 0) lds_st32(&ctl->wp, wp);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) lds_st32(&ctl->wp_epoch, pm >> 16);
            }
        }
        const uint32_t mode = pmode & 0xffffu;
        if (mode) {
#ifdef JTK_MCMC_STATS
            const unsigned long long tq0 = __builtin_readcyclecounter();
#endif
            // a start at q needs draws up to q + 63: the last positions wait for the next superblock
            bool parsed = false;
            while ((int32_t)(wr - (wp + 64)) >= 0) {
                if (mode == PM_K2) {
                    if ((int32_t)(wr - (wp + 3 * PKEEP + 64)) >= 0) {
                        producer_parse<4>(ring, rec, wp, parse_n, lane, seg_log);
                        wp += 4 * PKEEP;
                    } else {
                        producer_parse<1>(ring, rec, wp, parse_n, lane, seg_log);
                        wp += PKEEP;
                    }
                } else {
                    const uint32_t keep = mode <= 4u ? 44u : 32u;  // K - 1 more rejection loops need more look-ahead
                    producer_parse_gen(ring, rec, wp, parse_n, mode, keep, lane, seg_log);
                    wp += keep;
                }
                parsed = true;
            }
            if (parsed) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) lds_st32(&ctl->wp, wp);
            }
#ifdef JTK_MCMC_STATS
            st_parse += __builtin_readcyclecounter() - tq0;
#endif
        }
        if ((int32_t)(wr + SBLK - (uint32_t)c) > (int32_t)RN) {
#ifdef JTK_MCMC_STATS
            st_sleeps++;
#endif
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
#ifdef JTK_MCMC_STATS
        const unsigned long long tp0 = __builtin_readcyclecounter();
#endif
        uint64_t *blk = ring + (wr & (RN - 1));
#pragma unroll 8
        for (uint32_t j = 0; j < SEG; j++) {
            const uint64_t m5 = (x.s1 << 2) + x.s1, rr = rotl64(m5, 7);
            lds_st64(&blk[j * 64 + ((lane + j) & 63)], (rr << 3) + rr);  // rotl(s1 * 5, 7) * 9, skewed: no bank conflicts
            xo_step(x);
        }
        wr += SBLK;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lds_st32(&ctl->wr, wr);
#ifdef JTK_MCMC_STATS
        const unsigned long long tp1 = __builtin_readcyclecounter();
        st_gen += tp1 - tp0;
#endif
#ifdef JTK_MCMC_STATS
        const unsigned long long tp2 = __builtin_readcyclecounter();
#endif
        xo_jump(x, seg_log);
#ifdef JTK_MCMC_STATS
        st_jump += __builtin_readcyclecounter() - tp2;
#endif
    }
}

// The Bernoulli test of `0f64 < diff || rng.gen_bool(diff.exp())` (:736) for a step that does draw:
// gen_bool compares the u64 draw v with p_int = floor(exp(diff) * 2^64).  The exact exp is only evaluated
// when an f32 estimate with a guard band cannot decide, so the decision is always the exact one.
// (out of line: the exact exp is the rare path and the chain is sensitive to its code size)
__device__ __attribute__((noinline)) bool bernoulli_exact(uint64_t v, double diff) {
    // f32 estimate first: u = v / 2^64 within 2^-24, pe = exp(diff) within ~1e-5 relative
    const float u = (float)(uint32_t)(v >> 40) * 0x1p-24f;
    const float pe = __expf((float)diff);
    const bool in_range = diff < -1e-3 && diff > -44.4;
    if (ubool(diff <= -44.4 || (in_range && u > pe * 1.001f + 3e-7f))) return false;  // exp(diff) * 2^64 < 1 => p_int == 0
    if (ubool(in_range && u < pe * 0.999f - 3e-7f)) return true;
    const double scaled = unif64(jtk_exp(diff)) * 18446744073709551616.0;
    return v < uni64(__double2ull_rz(scaled));
}

// Neighbour-lane reads that stay off the LDS crossbar (a ds_bpermute round trip costs a lone wave ~100 cycles).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_next_lane(double v) { return dpp_f64<0x134>(v); }  // lane l <- lane l+1 (wave_rol:1)
__device__ __forceinline__ double from_prev_lane(double v) { return dpp_f64<0x13C>(v); }  // lane l <- lane l-1 (wave_ror:1)
__device__ __forceinline__ double wave_sum_f64(double v) {  // order-free: for estimates only
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);  // row_half_mirror
    v += dpp_f64<0x140>(v);  // row_mirror: every lane of a 16-lane row holds the row sum
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// mcmc_with_filter (:704-762), generic in K, one proposal per iteration.  m.assign holds the k-means labels on entry, the
// best-seen labels on exit.  This is the chain of mcmc_kernel_huge: pile-ups of more than JTK_MAX_PILEUP reads, or whose work
// area exceeds a CU's LDS -- there the per-read arrays of `m` point into a GLOBAL-memory workspace (every access below goes
// through generic pointers), the 10-bit read indices of the table-driven chains do not apply, and speed is not the point:
// clustering_on_pileup (local_clustering/mod.rs:86-123) takes any depth, so this library does too.
template <int K, bool SMALL>
__device__ __forceinline__ double mcmc_chain(const Lds &m, uint32_t n, uint32_t D, double cov, Rng &rng, uint32_t lane) {
    // size_to_lk[x] = max_{c=1..K} poisson_lk(x, cov*c)
    LaneTab size_to_lk;
#pragma unroll
    for (int r = 0; r < (SMALL ? 1 : 4); r++) {
        const uint32_t x = lane + 64 * r;
        double mx = -__builtin_inf();
        if (x <= n)
            for (int c = 1; c <= K; c++) {
                const double lam = cov * (double)c;
                mx = jtk_fmax(mx, (double)x * jtk_log(lam) - lam - m.lfact[x]);
            }
        size_to_lk.v[r] = mx;
    }
    // Pile-ups of more than 255 reads (high copy numbers: 8 copies x 40 reads) do not fit the four-register tables:
    // sizes and labels then live in LDS (m.size_to_lk, m.assign in place, m.argmax), one extra round trip per look-up.
    const bool big = !SMALL && n > 255u;
    if (big) {
        for (uint32_t x = lane; x <= n; x += 64) {
            double mx = -__builtin_inf();
            for (int c = 1; c <= K; c++) {
                const double lam = cov * (double)c;
                mx = jtk_fmax(mx, (double)x * jtk_log(lam) - lam - m.lfact[x]);
            }
            m.size_to_lk[x] = mx;
        }
        wsync();
    }
    auto size_lk = [&](uint32_t x) -> double { return big ? unif64(m.size_to_lk[x]) : tab_get<SMALL>(size_to_lk, x); };
    // ---- initial LKCounts in the reference's order (reads outer)
    double tg[K];
    int np[K], w[K], cl[K];
#pragma unroll
    for (int c = 0; c < K; c++) {
        tg[c] = 0.0;
        np[c] = 0;
        w[c] = 0;
        cl[c] = 0;
    }
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = uni(m.assign[i]);
        Elem el = {0.0, 0, 0};
        if (lane < D) el = elem_of(m.data[i * D + lane]);
#pragma unroll
        for (int cc = 0; cc < K; cc++)
            if ((uint32_t)cc == c) {
                tg[cc] += el.x;
                np[cc] += el.dp;
                w[cc] += el.pw;
                cl[cc]++;
            }
    }
    int totp = 0;  // reads with a positive value in this column: sum_c num_pos[c], constant along the chain
    unsigned long long posm[K], infm[K];
    const unsigned long long colm = D >= 64 ? ~0ull : ((1ull << D) - 1ull);
#pragma unroll
    for (int c = 0; c < K; c++) {
        totp += np[c];
        posm[c] = __ballot(0.0 < tg[c]) & colm;
        infm[c] = __ballot(w[c] > 0);
    }
    LaneLabels assign, argmax;
#pragma unroll
    for (int r = 0; r < (SMALL ? 1 : 4); r++) {
        const uint32_t i = lane + 64 * r;
        assign.v[r] = i < n ? (int)m.assign[i] : 0;
        argmax.v[r] = assign.v[r];
    }
    if (big) {
        for (uint32_t i = lane; i < n; i += 64) m.argmax[i] = m.assign[i];
        wsync();
    }
    // get_lk (:785-795) on a tentative state: size terms first, then clusters outer / columns inner, left to
    // right; exactly-zero terms (unused column or total_gain <= 0) leave the f64 sum unchanged and are skipped.
    auto get_lk = [&](const double *T, const int *P, const int *cls, const unsigned long long *pm,
                      const unsigned long long *im) -> double {
        double S = 0.0;
#pragma unroll
        for (int c = 0; c < K; c++) S += size_lk((uint32_t)cls[c]);
        int in_use = 0;
        unsigned long long anym = 0;
#pragma unroll
        for (int c = 0; c < K; c++) {
            in_use += (0.0 < T[c]) ? P[c] : 0;
            anym |= pm[c] & im[c];  // some cluster is_informative (:818-822) on this column
        }
        // get_used_columns (:847-869): informative somewhere, and 2 * pos_in_neg < pos_in_use
        const unsigned long long usedm = __ballot(3 * in_use > 2 * totp) & anym;
#pragma unroll
        for (int c = 0; c < K; c++) {
            unsigned long long mm = usedm & pm[c];
            while (mm) {
                const uint32_t d = (uint32_t)__builtin_ctzll(mm);
                mm &= mm - 1;
                S += readlane_f64(T[c], d);
            }
        }
        return S;
    };
    // The same quantity without the ordering (any order of the same terms: off by ~1e-12 at most).  The ordered sum
    // costs a v_readlane + a dependent add per term; this costs one cross-lane reduction, and it is enough to see
    // that a proposal is certainly rejected -- which > 96% of them are.
    // (its size terms come from three small per-cluster tables -- the entry of the current size, of one read less and
    // of one read more -- kept up to date on the rare accepts: no table look-up per proposal)
    double sz0[K], szm[K], szp[K];
    auto size_terms = [&](int c) {
        const uint32_t x = (uint32_t)cl[c];
        sz0[c] = size_lk(x);
        szm[c] = x > 0 ? size_lk(x - 1) : 0.0;
        szp[c] = x < n ? size_lk(x + 1) : 0.0;
    };
#pragma unroll
    for (int c = 0; c < K; c++) size_terms(c);
    auto approx_lk = [&](const double *T, const int *P, uint32_t from, uint32_t to, const unsigned long long *pm,
                         const unsigned long long *im) -> double {
        double S = 0.0;
#pragma unroll
        for (int c = 0; c < K; c++) S += (uint32_t)c == from ? szm[c] : ((uint32_t)c == to ? szp[c] : sz0[c]);
        int in_use = 0;
        unsigned long long anym = 0;
#pragma unroll
        for (int c = 0; c < K; c++) {
            in_use += (0.0 < T[c]) ? P[c] : 0;
            anym |= pm[c] & im[c];
        }
        const unsigned long long usedm = __ballot(3 * in_use > 2 * totp) & anym;
        const bool used = (usedm >> lane) & 1ull;
        double loc = 0.0;
#pragma unroll
        for (int c = 0; c < K; c++) loc += (used && 0.0 < T[c]) ? T[c] : 0.0;
        return S + wave_sum_f64(loc);
    };
    double lk = get_lk(tg, np, cl, posm, infm);
    double max = lk;
    const uint32_t total = 2000u * n;
#ifdef JTK_MCMC_STATS
    unsigned long long gs[6] = {0, 0, 0, 0, 0, 0};
#define GS_MARK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); gs[k] += now_ - gs_t; gs_t = now_; }
#else
#define GS_MARK(k)
#endif
    // Proposals are parsed from the 64-draw register window, not draw by draw.  A proposal is gen_range(0..n) -- the
    // first draw at or after its start whose widening product passes the zone test -- and then, for i = 1..K-1,
    // gen_index(i) on the upper halves of the following draws, each with its own zone test; the pick is the last i
    // whose index came out 0 (choose_pos).  Which draws pass which test, and which give index 0, depends on the draws
    // only: one ballot each per window, after which a proposal is a few scalar shift / find-first-set steps instead of
    // ~6 rejection loops on values that have to cross from the vector to the scalar side one at a time.
    uint32_t wp_base = 0xfffffff0u, wp_hi = 0;
    unsigned long long wp_ok0 = 0, wp_ok[K], wp_z[K];
#pragma unroll
    for (int i = 0; i < K; i++) wp_ok[i] = wp_z[i] = 0;
    const uint64_t zone_n = ((uint64_t)n << __clzll((long long)n)) - 1;
    for (uint32_t t = 0; t < total; t++) {
#ifdef JTK_MCMC_STATS
        unsigned long long gs_t = __builtin_readcyclecounter();
#endif
        uint32_t idx = 0, pos = 0;
        {
            uint32_t off = rng.pos - rng.win_base;
            if (off >= 40u) {  // keep 24 draws of look-ahead: reload the window at the current position
                rng_refill(rng);
                off = 0;
            }
            if (wp_base != rng.win_base) {
                const uint64_t v = rng.win;
                const uint32_t v32 = (uint32_t)(v >> 32);
                wp_hi = (uint32_t)__umul64hi(v, (uint64_t)n);
                wp_ok0 = __ballot(v * (uint64_t)n <= zone_n);
#pragma unroll
                for (int i = 1; i < K; i++) {
                    const uint32_t zone = ((uint32_t)i << __builtin_clz((uint32_t)i)) - 1u;
                    const uint64_t mi = (uint64_t)v32 * (uint32_t)i;
                    wp_ok[i] = __ballot((uint32_t)mi <= zone);
                    wp_z[i] = __ballot((uint32_t)(mi >> 32) == 0u);
                }
                wp_base = rng.win_base;
            }
            const unsigned long long m0 = wp_ok0 >> off;
            bool good = m0 != 0ull;
            const uint32_t p0 = off + (uint32_t)__builtin_ctzll(m0 | (1ull << 63));
            uint32_t q = p0;
#pragma unroll
            for (int i = 1; i < K; i++) {
                const unsigned long long mm = (good && q < 63u) ? wp_ok[i] >> (q + 1u) : 0ull;
                good = good && mm != 0ull;
                q = (q + 1u + (uint32_t)__builtin_ctzll(mm | (1ull << 63))) & 63u;
                if ((wp_z[i] >> q) & 1ull) pos = (uint32_t)i - 1u;
            }
            if (good) {
                idx = (uint32_t)__builtin_amdgcn_readlane((int)wp_hi, (int)p0);
                rng.pos = rng.win_base + q + 1u;
            } else {  // the proposal runs past the window: draw by draw
                idx = (uint32_t)gen_range_usize(rng, n);
                pos = choose_pos(rng, K);
            }
        }
        const uint32_t old = big ? uni((uint32_t)m.assign[idx]) : lab_get<SMALL>(assign, idx);
        const uint32_t nw = pos < old ? pos : pos + 1;
        GS_MARK(0);
        Elem el = {0.0, 0, 0};
        if (lane < D) el = elem_of(m.data[idx * D + lane]);
        // ---- tentative flip (:764-783): only the two touched clusters change
        double T[K];
        int P[K], W[K], ncl[K];
        unsigned long long npm[K], nim[K];
#pragma unroll
        for (int c = 0; c < K; c++) {
            const bool o = (uint32_t)c == old, a = (uint32_t)c == nw;
            T[c] = tg[c];
            P[c] = np[c];
            W[c] = w[c];
            ncl[c] = cl[c];
            npm[c] = posm[c];
            nim[c] = infm[c];
            if (o) {
                T[c] = tg[c] - el.x;
                P[c] = np[c] - el.dp;
                W[c] = w[c] - el.pw;
                ncl[c] = cl[c] - 1;
            }
            if (a) {
                T[c] = tg[c] + el.x;
                P[c] = np[c] + el.dp;
                W[c] = w[c] + el.pw;
                ncl[c] = cl[c] + 1;
            }
            if (o || a) {
                npm[c] = __ballot(0.0 < T[c]) & colm;
                nim[c] = __ballot(W[c] > 0);
            }
        }
        // estimate first: if proposed - lk is below -1e-3 the step certainly draws, and the draw usually settles it
        double proposed = 0.0;
        bool accept = false, decided = false, have_v = false;
        uint64_t v = 0;
        GS_MARK(1);
        const double dA = unif64(approx_lk(T, P, old, nw, npm, nim) - lk);
        GS_MARK(2);
        if (ubool(dA < -1e-3)) {
            v = next_u64(rng);
            have_v = true;
            const float u = (float)(uint32_t)(v >> 40) * 0x1p-24f;  // v / 2^64 within 2^-24
            decided = ubool(dA <= -44.5 || u > __expf((float)dA) * 1.001f + 3e-7f);  // certainly rejected
        }
        GS_MARK(3);
        if (!decided) {
#ifdef JTK_MCMC_STATS
            gs[5]++;
#endif
            proposed = get_lk(T, P, ncl, npm, nim);
            const double diff = unif64(proposed - lk);
            // `0f64 < diff || rng.gen_bool(diff.exp())` (:736): gen_bool(1.0) draws nothing, and exp(diff) == 1.0
            // exactly when diff >= -2^-54 (never the case when the estimate was below -1e-3)
            accept = true;
            if (!ubool(diff >= -0x1p-54)) accept = bernoulli_exact(have_v ? v : next_u64(rng), diff);
        }
        if (accept) {
#pragma unroll
            for (int c = 0; c < K; c++) {
                tg[c] = T[c];
                np[c] = P[c];
                w[c] = W[c];
                cl[c] = ncl[c];
                posm[c] = npm[c];
                infm[c] = nim[c];
                if ((uint32_t)c == old || (uint32_t)c == nw) size_terms(c);
            }
            if (big) {
                if (lane == 0) m.assign[idx] = (uint8_t)nw;
                wsync();
            } else {
                lab_set<SMALL>(assign, idx, nw, lane);
            }
            lk = proposed;
            if (ubool(max < lk)) {
                max = proposed;
                argmax = assign;
                if (big) {
                    for (uint32_t i = lane; i < n; i += 64) m.argmax[i] = m.assign[i];
                    wsync();
                }
            }
        } else {
            // flip back (:746): the reference re-adds / re-subtracts, which leaves rounding residue
#pragma unroll
            for (int c = 0; c < K; c++) {
                if ((uint32_t)c == old) tg[c] = T[c] + el.x;
                if ((uint32_t)c == nw) tg[c] = T[c] - el.x;
            }
        }
        GS_MARK(4);
    }
#ifdef JTK_MCMC_STATS
    if (lane == 0)
        printf("GENSTAT K %d n %u D %u steps %u draws %llu flip %llu approx %llu decide %llu tail %llu exact %llu\n", K, n, D,
               total, gs[0], gs[1], gs[2], gs[3], gs[4], gs[5]);
#endif
    wsync();
    if (big) {
        for (uint32_t i = lane; i < n; i += 64) m.assign[i] = m.argmax[i];
    } else {
#pragma unroll
        for (int r = 0; r < (SMALL ? 1 : 4); r++) {
            const uint32_t i = lane + 64 * r;
            if (i < n) m.assign[i] = (uint8_t)argmax.v[r];
        }
    }
    wsync();
    return max;
}

// The rejection threshold from the order-free estimate dA of proposed - lk.  u is the Bernoulli draw truncated to
// 19 bits (so the true uniform is < u + 2^-19); exp in f32 is good to ~1e-5 relative: 1.001 and 1.3e-6 cover both.
__device__ __forceinline__ float reject_threshold(double dA, bool pert) {
    float thr = 2.0f;  // cannot tell: the proposal becomes an event
    if (!pert && dA < -1e-3) thr = dA <= -44.5 ? -1.0f : __expf((float)dA) * 1.001f + 1.3e-6f;
    return thr;
}
// LDS accessors for the tables of the table-driven chains (generic pointers would make these flat accesses; structs
// travel as 16-byte vectors: one ds_read_b128 / ds_write_b128 each)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_c_u32x4;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) const double lds_c_f64;
typedef __attribute__((address_space(3))) float lds_f32;
__device__ __forceinline__ double lo_f64(u32x4 v) { return jtk_bits_f64(((uint64_t)v.y << 32) | v.x); }
__device__ __forceinline__ double hi_f64(u32x4 v) { return jtk_bits_f64(((uint64_t)v.w << 32) | v.z); }
__device__ __forceinline__ void lds_store_sz(SzEnt *p, const SzEnt &e) {
    lds_u32x4 *q = (lds_u32x4 *)p;
    const uint64_t a = jtk_f64_bits(e.rem), b = jtk_f64_bits(e.add);
    u32x4 v, w;
    v.x = (uint32_t)a;
    v.y = (uint32_t)(a >> 32);
    v.z = (uint32_t)b;
    v.w = (uint32_t)(b >> 32);
    w.x = e.nr;
    w.y = e.um;
    w.z = w.w = 0;
    q[0] = v;
    q[1] = w;
}
__device__ __forceinline__ SzEnt lds_load_sz(const SzEnt *p) {
    lds_c_u32x4 *q = (lds_c_u32x4 *)p;
    const u32x4 a = q[0];
    SzEnt e;
    e.rem = lo_f64(a);
    e.add = hi_f64(a);
    e.nr = ((__attribute__((address_space(3))) const uint32_t *)p)[4];  // byte 16
    e.um = ((__attribute__((address_space(3))) const uint32_t *)p)[5];
    e.pad[0] = e.pad[1] = 0;
    return e;
}

// ------------------------------------------------------------------------------------------------------
// The table-driven chain for any K (mcmc_chain_tab): K > 2, and the diploid pile-ups the fast path below does not take.
//
// As in the diploid chain, more than 96 % of the proposals are rejected, and the fate of "move read i from cluster a to
// cluster b" is a function of the state.  The producer wave has parsed the proposal that WOULD start at every stream
// position into a record (producer_parse_gen); for a window of 64 records the consumer evaluates, one proposal per lane,
// a REJECTION THRESHOLD and then steps from proposal to proposal with one v_readlane each:
//  * certainly rejected (the uniform behind its Bernoulli draw exceeds the threshold): the step is the reference's
//    flip + flip-back on the two touched clusters' sums -- (tg - x) + x and (tg + x) - x, rounding residue included --
//    and nothing else;
//  * anything else is an EVENT: one exact step with the reference's arithmetic (ordered left-to-right get_lk, exact exp
//    only if the guarded f32 test cannot decide), exactly as mcmc_chain does it.
// The threshold comes from an estimate of proposed - lk that is SEPARABLE: as long as the move flips no `0 < total_gain`
// and no column's used / unused status, get_lk changes by  s[b][i] - s[a][i]  (s[c][i] = sum of x[i][d] over the columns d
// that are used and where cluster c has a positive sum: LDS, rebuilt only when that column set changes) plus two size
// terms.  Whether a move can flip anything is certified per column with margins that hold for EVERY read: |total_gain|
// above the column's largest |x|, 3 pos_in_use - 2 total_pos away from 0 by more than one read, an informative cluster that
// stays informative under any +-7 change of its counter.  Columns that fail are collected in per-cluster / global bit
// masks; a proposal whose read has a non-zero value in such a column is never classified (it becomes an event), and
// neither is any proposal while a sum with counts behind it is within 1e-6 of zero (rounding residues, which move sums by
// ulps, could flip its sign).  Thresholds carry a 1e-3 guard band; the masks and s are republished at every accept and at
// least every 65,536 steps.
// Bit-identical to the one-step-at-a-time chain by construction; checked against the oracle.
struct GenWindow {
    uint32_t base;
    uint32_t nxt;   // per lane: window offset of the following proposal (Bernoulli draw taken), 255 = not in this window
    uint32_t ip;    // per lane: read index | pick << 10
    float u;        // per lane: the draw its Bernoulli test compares, / 2^64, truncated to 13 bits
};
__device__ __forceinline__ void gwindow_load(GenWindow &wd, Rng &rng, uint32_t base, uint32_t lane) {
    rng.pos = base;
    rng_release(rng, lane);
    rng_wait_rec(rng, base + 64);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wd.base = base;
    const uint32_t r = lds_ld32(&rng.rec[(base + lane) & (RN_OF(rng.seg_log) - 1)]);
    const uint32_t len = (r >> 13) & 63u;
    wd.ip = r & 0x1fffu;
    wd.nxt = (len != 0 && lane + len < 64) ? lane + len : 255u;
    wd.u = (float)(r >> 19) * 0x1p-13f;
}
typedef __attribute__((address_space(3))) const volatile uint8_t lds_cvu8;
__device__ __forceinline__ double lds_ld_f64(const double *p) { return jtk_bits_f64(lds_ld64(reinterpret_cast<const uint64_t *>(p))); }
// A pointer into LDS that reached this function through memory (a struct passed by reference, an argument register of an
// out-of-line call) looks divergent to the compiler: every use becomes a flat access with a null check and every branch
// on a value loaded through it an exec-mask region.  Rebuilt from its wave-uniform 32-bit LDS offset it is a scalar.
template <typename T>
__device__ __forceinline__ T *lds_uni(T *p) {
    typedef __attribute__((address_space(3))) char lds_char;
    const uint32_t off = uni((uint32_t)(uintptr_t)(lds_char *)const_cast<typename std::remove_const<T>::type *>(p));
    return (T *)(lds_char *)(uintptr_t)off;
}

template <int K>
__device__ __attribute__((noinline)) double mcmc_chain_tab(LdsShape shape, uint32_t n_in, uint32_t D_in, double cov_in,
                                                           Rng *rng_io, uint32_t lane) {
    // everything that steers control flow or addresses LDS is made provably wave-uniform first (see lds_uni)
    const uint32_t n = uni(n_in), D = uni(D_in);
    const double cov = unif64(cov_in);
    const Lds m = lds_carve(shape);
    Rng rng;
    rng.pos = uni(rng_io->pos);
    rng.wr_seen = uni(rng_io->wr_seen);
    rng.wp_seen = uni(rng_io->wp_seen);
    rng.win_base = uni(rng_io->win_base);
    rng.seg_log = uni(rng_io->seg_log);
    rng.pmode = uni(rng_io->pmode);
    rng.win = rng_io->win;
#ifdef JTK_MCMC_STATS
    rng.waits = rng_io->waits;
#endif
    rng.ctl = m.ctl;
    rng.ring = m.ring;
    rng.rec = m.rec;
    const bool small = n <= 63u, big = n > 255u;
    // size_to_lk[x] = max_{c=1..K} poisson_lk(x, cov*c), in LDS whatever n is (round 4: see size_lk below)
    {
        for (uint32_t x = lane; x <= n; x += 64) {
            double mx = -__builtin_inf();
            for (int c = 1; c <= K; c++) {
                const double lam = cov * (double)c;
                mx = jtk_fmax(mx, (double)x * jtk_log(lam) - lam - m.lfact[x]);
            }
            m.size_to_lk[x] = mx;
        }
        wsync();
    }
    // One LDS load, no branch.  Up to round 4 this read the table from its 4 registers for n <= 255 (`tab_get`: a switch on
    // x >> 6 around a pair of v_readlane): nine look-ups per accepted move and K per get_lk, ~8 branches each -- and a taken
    // branch costs a lone wave ~20 cycles: 2,200 -> 730 cycles for the state + size terms of an accept, 1,040 -> 480 for
    // get_lk (K = 2, 160 reads; profiles/r04_tab_event_breakdown.txt).
    auto size_lk = [&](uint32_t x) -> double { return unif64(m.size_to_lk[x]); };
    (void)small;
    (void)big;
    // ---- initial LKCounts in the reference's order (reads outer); lane = column
    double tg[K];
    int np[K], w[K], cl[K];
#pragma unroll
    for (int c = 0; c < K; c++) {
        tg[c] = 0.0;
        np[c] = 0;
        w[c] = 0;
        cl[c] = 0;
    }
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = uni(m.assign[i]);
        Elem el = {0.0, 0, 0};
        if (lane < D) el = elem_of(m.data[i * D + lane]);
#pragma unroll
        for (int cc = 0; cc < K; cc++)
            if ((uint32_t)cc == c) {
                tg[cc] += el.x;
                np[cc] += el.dp;
                w[cc] += el.pw;
                cl[cc]++;
            }
    }
    int totp = 0;
    const unsigned long long colm = D >= 64 ? ~0ull : ((1ull << D) - 1ull);
#pragma unroll
    for (int c = 0; c < K; c++) totp += np[c];
    // labels live in LDS (m.assign, with the best-seen copy in m.argmax): the walk reads a proposal's cluster from its
    // hop word, so only events and the threshold build look labels up
    for (uint32_t i = lane; i < n; i += 64) m.argmax[i] = m.assign[i];
    wsync();
    auto label_of = [&](uint32_t i) -> uint32_t { return uni((uint32_t) * (lds_cvu8 *)(m.assign + i)); };
    // get_lk (:785-795) on a (tentative) state: size terms first, then clusters outer / columns inner, left to right;
    // exactly-zero terms leave the f64 sum unchanged and are skipped
    auto get_lk = [&](const double *T, const int *P, const int *Wt, const int *cls) -> double {
        double S = 0.0;
#pragma unroll
        for (int c = 0; c < K; c++) S += size_lk((uint32_t)cls[c]);
        int in_use = 0;
        unsigned long long anym = 0, pm[K];
#pragma unroll
        for (int c = 0; c < K; c++) {
            pm[c] = __ballot(0.0 < T[c]) & colm;
            in_use += (0.0 < T[c]) ? P[c] : 0;
            anym |= pm[c] & __ballot(Wt[c] > 0);  // some cluster is_informative (:818-822) on this column
        }
        const unsigned long long usedm = __ballot(3 * in_use > 2 * totp) & anym;  // get_used_columns (:847-869)
#pragma unroll
        for (int c = 0; c < K; c++) {
            unsigned long long mm = usedm & pm[c];
            while (mm) {
                const uint32_t d = (uint32_t)__builtin_ctzll(mm);
                mm &= mm - 1;
                S += readlane_f64(T[c], d);
            }
        }
        return S;
    };
    // (the size terms of the K clusters -- size_to_lk of the size and of its two neighbours -- are looked up where they are
    // used, in publish(): kept in registers across the chain they were 3 K wave-uniform doubles, i.e. 6 K of the ~100 scalar
    // registers, and the chain's loop spilled scalars around every event)
    double lk = get_lk(tg, np, w, cl);
    double max = lk;
    // ---- thresholds (see the header comment).  Per-column constants first: the largest |x| and, per read, the columns
    //      with a non-zero value.
    const uint32_t npad = m.npad;
    typedef __attribute__((address_space(3))) const double lds_cd;
    typedef __attribute__((address_space(3))) double lds_d;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    lds_cd *const data_l = (lds_cd *)m.data;  // 32-bit LDS addressing for the hot gathers
    lds_d *const stab_l = (lds_d *)m.stab;
    lds_u32 *const nz_l = (lds_u32 *)m.nz;
    double xmax = 0.0;  // lane = column
    for (uint32_t i = 0; i < n; i++) {
        const double x = lane < D ? data_l[i * D + lane] : 0.0;
        xmax = fabs(x) > xmax ? fabs(x) : xmax;
    }
    for (uint32_t i = lane; i < n; i += 64) {
        uint32_t z = 0;
        for (uint32_t d = 0; d < D; d++) z |= data_l[i * D + d] != 0.0 ? 1u << d : 0u;
        nz_l[i] = z;
    }
    uint32_t umask[K];  // columns cluster c is paid for: used and total_gain > 0 (what s[c][.] is summed over)
#pragma unroll
    for (int c = 0; c < K; c++) umask[c] = 0xffffffffu;  // "never built"
    uint32_t nrcol = 0;   // columns whose used / unused status a single move could flip
    uint32_t nrun = 0;    // nrcol | every cluster's uncertified columns
    bool fragile = false; // some sum with counts behind it is within 1e-6 of zero
    double C0 = 0.0;      // (order-free get_lk of the current state) - lk: what every estimate starts from
    auto publish = [&]() {
        int IU = 0, AN = 0, RB = 0;
        bool alloff = true, frag = false;
        double G = 0.0;
        uint32_t nr[K], pm[K];
        double sz0[K], szm[K], szp[K];  // (the same value in every lane: LDS broadcast reads)
#pragma unroll
        for (int c = 0; c < K; c++) {
            const uint32_t x = (uint32_t)cl[c];
            sz0[c] = m.size_to_lk[x];
            szm[c] = x > 0 ? m.size_to_lk[x - 1] : 0.0;
            szp[c] = x < n ? m.size_to_lk[x + 1] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < K; c++) {
            const bool pos = 0.0 < tg[c];
            IU += pos ? np[c] : 0;
            AN += (pos && w[c] > 0) ? 1 : 0;
            RB += (pos && w[c] > 7) ? 1 : 0;
            alloff = alloff && (!pos || w[c] <= -7);
            frag = frag || (fabs(tg[c]) < 1e-6 && (np[c] != 0 || w[c] > 0));
            pm[c] = (uint32_t)(__ballot(pos) & colm);
            // `0 < total_gain` of this cluster cannot flip under any single move iff the sum clears the column's largest |x|
            nr[c] = (uint32_t)(__ballot(!(fabs(tg[c]) > xmax + 1e-6)) & colm);
        }
        if (lane < D) {  // the exact state, for the columns a proposal is not certified on (see hop_words)
#pragma unroll
            for (int c = 0; c < K; c++) {
                const uint64_t tb = jtk_f64_bits(tg[c]);
                u32x4 e;
                e.x = (uint32_t)tb;
                e.y = (uint32_t)(tb >> 32);
                e.z = (uint32_t)np[c];
                e.w = (uint32_t)w[c];
                ((lds_u32x4 *)m.st)[lane * K + c] = e;
            }
            u32x4 e;
            e.x = (uint32_t)IU;
            e.y = (uint32_t)AN;
            e.z = (uint32_t)totp;
            e.w = 0;
            ((lds_u32x4 *)m.col)[lane] = e;
        }
        const int v = 3 * IU - 2 * totp;  // used needs v >= 1; one move changes 3 IU by at most 3
        const bool iu_rob = v >= 4 || v <= -3;
        const bool an_rob = RB >= 1 || alloff;  // an informative cluster that stays one, or none that could become one
        const bool used = AN > 0 && v >= 1;
        nrcol = (uint32_t)(__ballot(!(iu_rob && an_rob)) & colm);
        nrun = nrcol;
#pragma unroll
        for (int c = 0; c < K; c++) nrun |= nr[c];
        fragile = __ballot(lane < D && frag) != 0ull;
        const uint32_t usedm = (uint32_t)(__ballot(used) & colm);
#pragma unroll
        for (int c = 0; c < K; c++) G += (used && 0.0 < tg[c]) ? tg[c] : 0.0;
        double S0 = 0.0;
#pragma unroll
        for (int c = 0; c < K; c++) S0 += sz0[c];
        C0 = unif64((S0 + wave_sum_f64(lane < D ? G : 0.0)) - lk);
#pragma unroll
        for (int c = 0; c < K; c++) {
            const uint32_t um = usedm & pm[c];
            if (um != umask[c]) {  // rare once the clusters have formed: rebuild s[c][.]
                umask[c] = um;
                for (uint32_t i = lane; i < n; i += 64) {
                    double sc = 0.0;
                    uint32_t mm = um;
                    while (mm) {
                        const uint32_t d = (uint32_t)__builtin_ctz(mm);
                        mm &= mm - 1;
                        sc += data_l[i * D + d];
                    }
                    stab_l[(uint32_t)c * npad + i] = sc;
                }
            }
            if (lane == 0) {
                SzEnt e;
                e.rem = szm[c] - sz0[c];
                e.add = szp[c] - sz0[c];
                e.nr = nr[c];
                e.um = um;
                e.pad[0] = e.pad[1] = 0;
                lds_store_sz(&m.sz[c], e);
            }
        }
        wsync();
    };
    // per window position: nxt (6 bits) | certainly rejected << 6 | in-window << 7 | read index << 8 | pick << 18 |
    // the read's current cluster << 21 | the cluster the proposal moves it to << 24
    const uint32_t n1 = n - 1;
    auto hop_words = [&](const GenWindow &wd) -> uint32_t {
        uint32_t idx = wd.ip & 1023u;
        uint32_t pick = wd.ip >> 10;
        const bool in = wd.nxt != 255u;
        idx = idx < n1 ? idx : n1;  // a position that is not a parsed proposal may hold anything
        pick = pick < (uint32_t)(K - 1) ? pick : 0u;
        const uint32_t old = *(lds_cvu8 *)(m.assign + idx);
        const uint32_t nw = pick < old ? pick : pick + 1u;
        const SzEnt ea = lds_load_sz(&m.sz[old]), eb = lds_load_sz(&m.sz[nw]);
        const uint32_t z = ((lds_cu32 *)nz_l)[idx];
        // columns this proposal is not certified on: there the change of get_lk is evaluated from the exact state
        const uint32_t F = z & (ea.nr | eb.nr | nrcol);
        bool cant = fragile;
        double corr = 0.0;
        uint32_t any = nrun;  // the columns some proposal could be uncertified on (wave-uniform)
        while (any) {
            const uint32_t d = (uint32_t)__builtin_ctz(any);
            any &= any - 1;
            if (!((F >> d) & 1u)) continue;
            const Elem el = elem_of(data_l[idx * D + d]);
            const u32x4 ce = ((lds_c_u32x4 *)m.col)[d];
            const u32x4 qa = ((lds_c_u32x4 *)m.st)[d * K + old], qb = ((lds_c_u32x4 *)m.st)[d * K + nw];
            const double Ta0 = lo_f64(qa), Tb0 = lo_f64(qb);
            const int Pa = (int)qa.z, Wa = (int)qa.w, Pb = (int)qb.z, Wb = (int)qb.w;
            const bool pa = 0.0 < Ta0, pb = 0.0 < Tb0;
            const double Ta = Ta0 - el.x, Tb = Tb0 + el.x;
            const int Pa2 = Pa - el.dp, Wa2 = Wa - el.pw, Pb2 = Pb + el.dp, Wb2 = Wb + el.pw;
            const bool pa2 = 0.0 < Ta, pb2 = 0.0 < Tb;
            const int IU = (int)ce.x, AN = (int)ce.y, TP = (int)ce.z;
            const bool used0 = AN > 0 && 3 * IU > 2 * TP;
            const int IU2 = IU - (pa ? Pa : 0) - (pb ? Pb : 0) + (pa2 ? Pa2 : 0) + (pb2 ? Pb2 : 0);
            const int AN2 = AN - ((pa && Wa > 0) ? 1 : 0) - ((pb && Wb > 0) ? 1 : 0) + ((pa2 && Wa2 > 0) ? 1 : 0) +
                            ((pb2 && Wb2 > 0) ? 1 : 0);
            const bool used2 = AN2 > 0 && 3 * IU2 > 2 * TP;
            // the two clusters' terms before and after; the other clusters' terms only matter if `used` flips
            const double t0 = used0 ? ((pa ? Ta0 : 0.0) + (pb ? Tb0 : 0.0)) : 0.0;
            const double t2 = used2 ? ((pa2 ? Ta : 0.0) + (pb2 ? Tb : 0.0)) : 0.0;
            if (used0 != used2) cant = true;  // (every other cluster's term switches too: rare, left to the exact step)
            // sums near zero with counts behind them: rounding residues could flip their sign
            cant = cant || (fabs(Ta) < 1e-6 && (Pa2 != 0 || Wa2 > 0)) || (fabs(Tb) < 1e-6 && (Pb2 != 0 || Wb2 > 0));
            // replace the separable contribution of this column by the exact one
            const double sep = (((eb.um >> d) & 1u) ? el.x : 0.0) - (((ea.um >> d) & 1u) ? el.x : 0.0);
            corr += (t2 - t0) - sep;
        }
        const double dA = (((stab_l[nw * npad + idx] - stab_l[old * npad + idx]) + corr) + (ea.rem + eb.add)) + C0;
        const float t = cant ? 2.0f : reject_threshold(dA, false);
        return (wd.nxt & 63u) | ((in && wd.u > t) ? 64u : 0u) | (in ? 128u : 0u) | (idx << 8) | (pick << 18) | (old << 21) | (nw << 24);
    };
#ifdef JTK_MCMC_STATS
    unsigned long long ts[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // fast, events, accepts, reloads, scalars, cyc rebuild, cyc event, residues, cyc window load, cyc hop words, uncertified columns
    const unsigned long long ts_t0 = __builtin_readcyclecounter();
#define TS_ADD(k, v) ts[k] += (v)
#else
#define TS_ADD(k, v)
#endif
    publish();
    const uint32_t total = 2000u * n;
    uint32_t t = 0, p = 0, since_rebuild = 0;
    uint32_t n_events = 0;  // (reported per chunk: jtk_lc_debug_chain_profile)
    GenWindow wd;
    gwindow_load(wd, rng, rng.pos, lane);
    uint32_t hopw = hop_words(wd);
    const uint32_t row_lane_addr = (uint32_t)(uintptr_t)data_l + (lane < D ? lane : 0u) * 8u;  // (the rejected steps' hand-issued loads)
    // K <= 4 clusters of <= 16 columns (every BASELINE shape): during the quiet part of the chain the K sums of a column travel in
    // ONE register, lane 16 c + d = cluster c / column d, so that a rejected step is one multiplication and two additions for all
    // clusters, its factor picked per lane (two compares) instead of per cluster (four scalar instructions each).  The exact step
    // and publish() keep their register per cluster: packed on the way into the first block after an event, unpacked before the
    // next event (2 K ds_bpermute each way, against ~4,000 cycles of event).
    const bool packable = K <= 4 && D <= 16u;
    const uint32_t grp = lane >> 4, col16 = lane & 15u;
    const uint32_t row_lane_addr_pk = (uint32_t)(uintptr_t)data_l + (col16 < D ? col16 : 0u) * 8u;
    double tgp = 0.0;
    bool is_packed = false;
    auto pack_sums = [&]() {
        tgp = 0.0;
#pragma unroll
        for (int c = 0; c < K; c++) {
            const double v = __shfl(tg[c], (int)col16, 64);  // (lanes >= D of tg[c] hold +0)
            tgp = grp == (uint32_t)c ? v : tgp;
        }
        is_packed = true;
    };
    auto unpack_sums = [&]() {
#pragma unroll
        for (int c = 0; c < K; c++) {
            const double v = __shfl(tgp, (int)(16u * (uint32_t)c + col16), 64);
            tg[c] = lane < D ? v : 0.0;
        }
        is_packed = false;
    };
    auto row_of = [&](uint32_t hvv) -> double {  // the column values of the read a hop word names (lanes >= D: 0.0)
        uint32_t i = (hvv >> 8) & 1023u;
        i = i < n1 ? i : n1;  // a word that is not a proposal may hold anything
        return lane < D ? data_l[i * D + lane] : 0.0;
    };
    for (;;) {
        // ---- the quiet part, a loop of its own: blocks of certainly rejected proposals and window moves follow one another
        //      without passing the event's code (whose many live values the compiler would otherwise merge at every back edge:
        //      a chain spends ~8 steps per window and ~70 steps per event at K = 3)
        uint32_t hv = 0;
        bool finished = false;
        for (;;) {
        if (t >= total) {
            finished = true;
            break;
        }
        hv = uni((uint32_t)__builtin_amdgcn_readlane((int)hopw, (int)p));
        if ((hv & 192u) == 192u && since_rebuild < 65536u) {
#ifdef JTK_MCMC_STATS
            const unsigned long long f_t0 = __builtin_readcyclecounter();
#endif
            // ---- certainly rejected proposals, one after the other: flip + flip back (:739,:746) on the two touched
            //      clusters and nothing else.  The next proposal's hop word and row are fetched before this one's
            //      arithmetic (an LDS round trip costs a lone wave ~100 cycles).  Round 5, from the ISA of the round-4 loop:
            //      (a) `x = xn` at the back edge made every step wait for the row it had just asked for -- the loop is unrolled
            //      twice over two row registers, a step waits for the OLDER load only; (b) 2 K scalar compare-and-branch pairs
            //      picked the two touched sums -- now every cluster's sum takes (s + m) - m with m = x * {-1, +1, 0} (a scalar
            //      factor): x * -1 and x * 1 are exact, (s + -x) - -x is (s - x) + x bit for bit, and m = +-0 leaves s as it is
            //      (no sum is ever -0: they grow from +0 by additions), so the bits are the reference's and nothing branches.
            uint32_t budget = total - t;
            if (budget > 65536u - since_rebuild) budget = 65536u - since_rebuild;
            uint32_t done = 0;
            auto rejected_step = [&](uint32_t hvv, double x, auto use_packed) {
                const uint32_t old = (hvv >> 21) & 7u, nw = (hvv >> 24) & 7u;
#ifdef JTK_MCMC_STATS
                bool st_res = false;
#endif
                if (decltype(use_packed)::value) {
                    const uint32_t hi = grp == old ? 0xBFF00000u : (grp == nw ? 0x3FF00000u : 0u);
                    const double mc = x * __hiloint2double((int)hi, 0);
#ifdef JTK_MCMC_STATS
                    TS_ADD(7, __ballot((tgp + mc) - mc != tgp) != 0ull ? 1 : 0);
#endif
                    tgp = (tgp + mc) - mc;
                    return;
                }
#pragma unroll
                for (int c = 0; c < K; c++) {
                    const uint32_t hi = (uint32_t)c == old ? 0xBFF00000u : ((uint32_t)c == nw ? 0x3FF00000u : 0u);
                    const double mc = x * __hiloint2double((int)hi, 0);
#ifdef JTK_MCMC_STATS
                    st_res = st_res || (tg[c] + mc) - mc != tg[c];
#endif
                    tg[c] = (tg[c] + mc) - mc;
                }
#ifdef JTK_MCMC_STATS
                TS_ADD(7, __ballot(st_res) != 0ull ? 1 : 0);  // rejected steps that leave a rounding residue in some sum
#endif
            };
            // The rows travel through hand-issued ds_read_b64 with hand-placed waits: the compiler's own scoreboard waits with
            // lgkmcnt(0) in this loop -- i.e. for the row it has just asked for as well -- where "all but the youngest load"
            // (lgkmcnt(1): LDS loads return in order) is what hides the round trip.  No other LDS access happens between the
            // first load and the drain behind the loop; lanes >= D read column 0 and keep their zero sums ((0 + m) - m == +0).
            auto row_issue = [&](uint32_t hvv, auto use_packed) -> double {
                const uint32_t i = (hvv >> 8) & 1023u;  // (hop_words clamps the index of a position that is not a proposal)
                double v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v)
                             : "v"((decltype(use_packed)::value ? row_lane_addr_pk : row_lane_addr) + i * (D * 8u)));
                return v;
            };
            // A window holds at most 21 proposals (three draws each at least): with 24 steps of budget left -- always, but at
            // the very end of a chain and once per 65,536 steps -- the block ends with the window and nothing counts steps
            // against the budget (three scalar instructions per step less).
            auto run_block = [&](auto watch_budget, auto use_packed) {
                double xa = row_issue(hv, use_packed), xb;
                for (;;) {
                    uint32_t pn = hv & 63u;
                    uint32_t hn = uni((uint32_t)__builtin_amdgcn_readlane((int)hopw, (int)pn));
                    xb = row_issue(hn, use_packed);
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(xa));  // the older of the two rows in flight
                    rejected_step(hv, xa, use_packed);
                    p = pn;
                    hv = hn;
                    done++;
                    if ((hv & 192u) != 192u) break;
                    if (decltype(watch_budget)::value && done >= budget) break;
                    pn = hv & 63u;
                    hn = uni((uint32_t)__builtin_amdgcn_readlane((int)hopw, (int)pn));
                    xa = row_issue(hn, use_packed);
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(xb));
                    rejected_step(hv, xb, use_packed);
                    p = pn;
                    hv = hn;
                    done++;
                    if ((hv & 192u) != 192u) break;
                    if (decltype(watch_budget)::value && done >= budget) break;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa), "+v"(xb));  // the row fetched for a step that did not run
            };
            if (packable) {
                if (!is_packed) pack_sums();
                if (budget >= 24u)
                    run_block(std::false_type(), std::true_type());
                else
                    run_block(std::true_type(), std::true_type());
            } else if (budget >= 24u) {
                run_block(std::false_type(), std::false_type());
            } else {
                run_block(std::true_type(), std::false_type());
            }
            t += done;
            since_rebuild += done;
            TS_ADD(0, done);
#ifdef JTK_MCMC_STATS
            ts[11] += __builtin_readcyclecounter() - f_t0;   // cycles inside the rejected-steps block
            ts[12] += 1;                                      // its entries
#endif
            continue;
        }
        if (!(hv & 128u) && p != 0) {  // the proposal does not end inside this window: move the window there
#ifdef JTK_MCMC_STATS
            const unsigned long long w_t0 = __builtin_readcyclecounter();
#endif
            gwindow_load(wd, rng, wd.base + p, lane);
#ifdef JTK_MCMC_STATS
            const unsigned long long w_t1 = __builtin_readcyclecounter();
#endif
            p = 0;
            hopw = hop_words(wd);
            TS_ADD(3, 1);
#ifdef JTK_MCMC_STATS
            ts[8] += w_t1 - w_t0;
            ts[9] += __builtin_readcyclecounter() - w_t1;
            ts[10] += (unsigned long long)__popc(nrun);
#endif
            continue;
        }
        break;
        }
        if (is_packed) unpack_sums();
        if (finished) break;
        uint32_t idx, pick, pos_v;
        bool reload = false;
        if (hv & 128u) {
            idx = (hv >> 8) & 1023u;
            pick = (hv >> 18) & 7u;
            pos_v = wd.base + (hv & 63u) - 1;
        } else {  // not even at the window start: the producer could not parse this one -- scalar draws
            TS_ADD(4, 1);
            rng.pos = wd.base;
            idx = (uint32_t)gen_range_usize(rng, n);
            pick = choose_pos(rng, K);
            pos_v = rng.pos;
            reload = true;
        }
        // ---- the event: one exact step (as mcmc_chain)
#ifdef JTK_MCMC_STATS
        const unsigned long long ev_t0 = __builtin_readcyclecounter();
#endif
        TS_ADD(1, 1);
        n_events++;
        const uint32_t old = label_of(idx);
        const uint32_t nw = pick < old ? pick : pick + 1;
        Elem el = {0.0, 0, 0};
        if (lane < D) el = elem_of(lds_ld_f64(&m.data[idx * D + lane]));
        double T[K];
        int P[K], W[K], ncl[K];
#pragma unroll
        for (int c = 0; c < K; c++) {
            const bool o = (uint32_t)c == old, a = (uint32_t)c == nw;
            T[c] = o ? tg[c] - el.x : (a ? tg[c] + el.x : tg[c]);
            P[c] = o ? np[c] - el.dp : (a ? np[c] + el.dp : np[c]);
            W[c] = o ? w[c] - el.pw : (a ? w[c] + el.pw : w[c]);
            ncl[c] = o ? cl[c] - 1 : (a ? cl[c] + 1 : cl[c]);
        }
        const double proposed = get_lk(T, P, W, ncl);
        const double diff = unif64(proposed - lk);
        // `0f64 < diff || rng.gen_bool(diff.exp())` (:736): gen_bool(1.0) draws nothing, and exp(diff) == 1.0 exactly
        // when diff >= -2^-54
        const bool no_draw = ubool(diff >= -0x1p-54);
        bool accept = true;
        if (!no_draw) {
            const float u = reload ? -1.0f : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wd.u), (int)p));
            const float pe = __expf((float)diff);
            const bool in_range = u >= 0.0f && diff < -1e-3 && diff > -44.4;
            if (ubool(diff <= -44.4 || (in_range && u > pe * 1.001f + 1.3e-6f))) {
                accept = false;
            } else if (!ubool(in_range && u + 0x1p-13f < pe * 0.999f - 3e-7f)) {
                rng_wait(rng, pos_v + 1);
                accept = ubool(bernoulli_exact(uni64(lds_ld64(&rng.ring[ring_slot(pos_v, rng.seg_log)])), diff));
            }
        }
        if (accept) {
#pragma unroll
            for (int c = 0; c < K; c++) {
                tg[c] = T[c];
                np[c] = P[c];
                w[c] = W[c];
                cl[c] = ncl[c];
            }
            if (lane == 0) m.assign[idx] = (uint8_t)nw;
            wsync();
            lk = proposed;
            if (ubool(max < lk)) {
                max = proposed;
                for (uint32_t i = lane; i < n; i += 64) m.argmax[i] = m.assign[i];
                wsync();
            }
        } else {
            // flip back (:746): the reference re-adds / re-subtracts, which leaves rounding residue
#pragma unroll
            for (int c = 0; c < K; c++) {
                if ((uint32_t)c == old) tg[c] = T[c] + el.x;
                if ((uint32_t)c == nw) tg[c] = T[c] - el.x;
            }
        }
        t++;
        since_rebuild++;
        const bool rebuilt = accept || since_rebuild >= 65536u;
        TS_ADD(2, accept ? 1 : 0);
        if (rebuilt) {
#ifdef JTK_MCMC_STATS
            const unsigned long long rb_t0 = __builtin_readcyclecounter();
#endif
            publish();
            since_rebuild = 0;
            TS_ADD(5, __builtin_readcyclecounter() - rb_t0);
        }
        TS_ADD(6, __builtin_readcyclecounter() - ev_t0);
        const uint32_t pos_next = no_draw ? pos_v : pos_v + 1;
        if (reload || pos_next - wd.base >= 64) {
            gwindow_load(wd, rng, pos_next, lane);
            p = 0;
            hopw = hop_words(wd);
        } else {
            p = pos_next - wd.base;
            if (rebuilt) hopw = hop_words(wd);
        }
    }
#ifdef JTK_MCMC_STATS
    if (lane == 0)
        printf("TABSTAT chunk %u K %d n %u D %u steps %u fast %llu events %llu accepts %llu reloads %llu scalars %llu cyc_rebuild %llu cyc_event %llu cyc_total %llu residues %llu cyc_wload %llu cyc_hopw %llu uncert %llu cyc_fast %llu fast_entries %llu\n",
               blockIdx.x, K, n, D, total, ts[0], ts[1], ts[2], ts[3], ts[4], ts[5], ts[6], __builtin_readcyclecounter() - ts_t0, ts[7], ts[8], ts[9], ts[10], ts[11], ts[12]);
#endif
#undef TS_ADD
    if (lane == 0) m.k2_stats[16] += n_events;
    rng.pos = wd.base + p;
    rng_release(rng, lane);
    wsync();
    for (uint32_t i = lane; i < n; i += 64) m.assign[i] = m.argmax[i];
    wsync();
    *rng_io = rng;
    return max;
}

// ------------------------------------------------------------------------------------------------------
// ---- the diploid chain (round 5).
//
// Lane i of the consumer holds READ i (and read 64 + i when NR == 2): its row, signed by the direction of its flip, and --
// rebuilt lane-parallel after every move of the state -- the EXACT likelihood of the state with that read flipped (get_lk's
// own left-to-right sum) and whether flip + flip-back would leave a rounding residue.  Both go to a 16-byte entry per read
// in LDS.  A window is 64 consecutive stream positions (lane l = position base + l): the producer's record of the proposal
// that WOULD start there (read index, length, 19 bits of its Bernoulli draw); per window the known bits of the draw become
// two thresholds in the log domain, and the hop word of a position follows from its read's entry: diff = proposed - lk below
// the one: certainly rejected, above the other: certainly accepted.  The walk follows the hop words over proposals that are
// certainly rejected and leave nothing behind; everything else is an event, settled from the hop word (the exact exp only
// inside the guard bands) and the read's lane.  The state (LKCount[c][d] of the two clusters) is wave-uniform and replicated
// in every lane: neither the event nor the re-evaluation needs a cross-lane operation.  Size-only moves (all-zero rows) are
// not a special case: the size terms are where every lane's sum starts.
//
// The hop word of window position l (one v_readlane per hop yields all of it):
//   bits 0..5 nxt[l] | 64 skip: inside the window, certainly rejected, no residue | 128 certainly accepted
//   | 256 accepted without a draw | 512 certainly rejected | 1024 not in this window | 2048 a rejected flip leaves a residue
#define HW_SKIP 64u
#define HW_ACC 128u
#define HW_NODRAW 256u
#define HW_REJ 512u
#define HW_OUT 1024u
#define HW_PERT 2048u
#define HW_CROSS 4096u   // (round 6) the proposal ends in the NEXT block of 64 positions: bits 0..5 are its end there
#define HW_SKIPX 8192u   // ... and is certainly rejected without a residue (== HW_SKIP << 7: never both)
typedef __attribute__((address_space(3))) const volatile double lds_cvf64;
typedef __attribute__((address_space(3))) const volatile u32x4_t lds_cvu32x4;
typedef __attribute__((address_space(3))) volatile u32x4_t lds_vu32x4;
__device__ __forceinline__ uint32_t lds_addr(const void *p) {  // LDS byte address (a generic pointer indexed per lane costs a
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;  // 64-bit add and a null check per access)
}
// Walks from window position p over skippable proposals; returns the number of steps taken (<= limit) and the hop word it
// stopped at.  Straight-line hops with forward exits: a taken branch costs a lone wave far more than the hop itself.
__device__ __forceinline__ uint32_t walk_rejected(uint32_t hopw, uint32_t &p, uint32_t limit, uint32_t &hv_out) {
    uint32_t hv = 0;
    if (limit >= 24) {  // a window holds at most 21 proposals: no need to watch the step budget
        uint32_t steps = 22;
#pragma unroll
        for (uint32_t k = 0; k < 22; k++) {
            hv = uni((uint32_t)__builtin_amdgcn_readlane((int)hopw, (int)p));
            if (!(hv & HW_SKIP)) {
                steps = k;
                break;
            }
            p = hv & 63u;
        }
        hv_out = hv;
        return steps;
    }
    uint32_t steps = 0;
    while (steps < limit) {
        hv = uni((uint32_t)__builtin_amdgcn_readlane((int)hopw, (int)p));
        if (!(hv & HW_SKIP)) break;
        p = hv & 63u;
        steps++;
    }
    hv_out = hv;
    return steps;
}
// One proposal taken with scalar draws (a start the producer could not parse): the read index and the stream
// position of the draw a Bernoulli test would compare.
__device__ __forceinline__ void scalar_proposal(Rng &rng, uint32_t start, uint32_t n, uint32_t &idx, uint32_t &pos_v) {
    rng.pos = start;
    idx = (uint32_t)gen_range_usize(rng, n);
    (void)gen_index(rng, 1);  // choose() over the single other cluster (pseudo_mcmc.rs:732)
    pos_v = rng.pos;
}

#ifdef JTK_MCMC_STATS
// counters live in scalar registers during the chain and are folded into LDS once per chain
#define ST_T0() unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_mk = 0; (void)st_mk; const unsigned long long st_t0 = __builtin_readcyclecounter()
#define ST_ADD(k) st_acc[k] += __builtin_readcyclecounter() - st_t0; if (lane == 0) { for (int q_ = 0; q_ < 16; q_++) m.k2_stats[q_] += st_acc[q_]; }
#define ST_CNT(k, v) st_acc[k] += (v)
// the pieces of an event: ST_MARK0 starts the stopwatch, ST_MARK(k) adds the time since the last mark to counter k
#define ST_MARK0() st_mk = __builtin_readcyclecounter()
#define ST_MARK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[k] += now_ - st_mk; st_mk = now_; }
#else
#define ST_T0()
#define ST_ADD(k)
#define ST_CNT(k, v)
#define ST_MARK0()
#define ST_MARK(k)
#endif

// Out of line on purpose: inlined into the kernel, the chain inherits the register pressure of everything that is
// live around it and spills scalar registers inside its loop (each reload is a v_readlane on the critical path).
struct K2Mem {
    const double *data;  // n x D likelihood gains
    const double *lfact;
    uint8_t *assign;
    double *tab;         // 16 n bytes: one entry per read (the k-means scratch fbuf + cum: idle during a chain)
    unsigned long long *k2_stats;
};
template <typename T>
__device__ __forceinline__ T *uni_ptr(T *p) { return reinterpret_cast<T *>((uintptr_t)uni64((uint64_t)(uintptr_t)p)); }
// NR: registers per per-read / per-size table: 1 serves n <= 63, 2 serves n <= 127 (read or size 64 r + lane)
template <int DMAX, int NR>
__device__ __attribute__((noinline)) double mcmc_chain_k2(K2Mem m_in, uint32_t n_in, uint32_t D_in, double cov_in, Rng *rng_io,
                                                          uint32_t lane) {
    // Arguments of an out-of-line function arrive in vector registers and the compiler then treats everything derived from
    // them -- the step counter, the window position, every branch of the event -- as divergent (exec-mask regions instead
    // of scalar branches): all of it is re-made wave-uniform here.
    const uint32_t n = uni(n_in), D = uni(D_in);
    const double cov = unif64(cov_in);
    const K2Mem m = {uni_ptr(m_in.data), uni_ptr(m_in.lfact), uni_ptr(m_in.assign), uni_ptr(m_in.tab), uni_ptr(m_in.k2_stats)};
    Rng rng = *rng_io;
    rng.pos = uni(rng.pos);
    rng.wr_seen = uni(rng.wr_seen);
    rng.wp_seen = uni(rng.wp_seen);
    rng.pmode = uni(rng.pmode);
    rng.win_base = uni(rng.win_base);
    rng.seg_log = uni(rng.seg_log);
    const uint32_t rn_mask = RN_OF(rng.seg_log) - 1u;
    rng.ctl = uni_ptr(rng.ctl);
    rng.ring = uni_ptr(rng.ring);
    rng.rec = uni_ptr(rng.rec);
    const uint32_t rec_lds = uni(lds_addr(rng.rec)), data_lds = uni(lds_addr(m.data)), tab_lds = uni(lds_addr(m.tab));
    // pair table: lane c0 holds (0.0 + size_to_lk[c0]) + size_to_lk[n - c0]   (get_lk :788)
    double pair_v[NR];
    auto tab64 = [&](const double *tab, uint32_t i) -> double {  // entry i of a per-lane table of NR registers
        return NR == 2 && i >= 64 ? readlane_f64(tab[NR - 1], i & 63u) : readlane_f64(tab[0], i & 63u);
    };
    auto bit128 = [&](const unsigned long long *mk, uint32_t i) -> bool {
        return ((NR == 2 && i >= 64 ? mk[NR - 1] : mk[0]) >> (i & 63u)) & 1ull;
    };
    {
        auto size_lk = [&](uint32_t x) {
            double mx = -__builtin_inf();
            for (int c = 1; c <= 2; c++) {
                const double lam = cov * (double)c;
                mx = jtk_fmax(mx, (double)x * jtk_log(lam) - lam - m.lfact[x]);
            }
            return mx;
        };
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const uint32_t c = lane + 64 * r, cc = c <= n ? c : n;
            pair_v[r] = (0.0 + size_lk(cc)) + size_lk(n - cc);
        }
    }
    // ---- exact state LKCount[c][d]; columns >= D are all-zero, never used and add +0.0.  The two counters travel
    //      packed: pk = num_pos + 65536 * (3*num_pos - 7*num_neg), so pk > 0xffff <=> the second one is positive.
    double tg0[DMAX], tg1[DMAX];
    int pk0[DMAX], pk1[DMAX], tp2[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; d++) {
        tg0[d] = tg1[d] = 0.0;
        pk0[d] = pk1[d] = tp2[d] = 0;
    }
    uint32_t c0 = 0;
    unsigned long long lab[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) lab[r] = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = uni(m.assign[i]);
#pragma unroll
        for (int d = 0; d < DMAX; d++) {
            Elem el = {0.0, 0, 0};
            if ((uint32_t)d < D) el = elem_of(unif64(m.data[i * D + d]));
            tp2[d] += 2 * el.dp;  // 2 x reads with a positive value in this column: constant along the chain
            if (c == 0) {
                tg0[d] += el.x;
                pk0[d] += el.dp + 65536 * el.pw;
            } else {
                tg1[d] += el.x;
                pk1[d] += el.dp + 65536 * el.pw;
            }
        }
        if (c == 0)
            c0++;
        else if (NR == 2 && i >= 64)
            lab[NR - 1] |= 1ull << (i & 63u);
        else
            lab[0] |= 1ull << (i & 63u);
    }
    // ---- the rows of reads lane, 64 + lane, signed by the direction of their flip: sx[r][d] is what cluster 0 would gain
    double sx[NR][DMAX];
    int spk[NR][DMAX];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const uint32_t ri = lane + 64 * r < n ? lane + 64 * r : 0;
        const bool a = bit128(lab, ri);
#pragma unroll
        for (int d = 0; d < DMAX; d++) {
            Elem el = {0.0, 0, 0};
            if ((uint32_t)d < D) el = elem_of(*(lds_cvf64 *)(uintptr_t)(data_lds + ((ri * D + (uint32_t)d) << 3)));
            const int kk = el.dp + 65536 * el.pw;
            sx[r][d] = a ? el.x : -el.x;
            spk[r][d] = a ? kk : -kk;
        }
    }
    wsync();
    auto pair_at = [&](uint32_t c) -> double { return tab64(pair_v, c <= n ? c : n); };
    // get_lk (:785-795) of the state in which a read with the signed row (x, k), sitting in cluster 1 iff `a`, is flipped --
    // exactly: the size terms, then clusters outer / columns inner, left to right -- and whether flip + flip-back (:746)
    // would leave a rounding residue in the sums
    double pair_up, pair_dn;
    auto flipped_lk = [&](const double *x, const int *k, bool a, double &S_out, bool &pert_out) {
        double t1[DMAX];
        double S = a ? pair_up : pair_dn;
        bool pert = false;
#pragma unroll
        for (int d = 0; d < DMAX; d++) {
            const double T0 = tg0[d] + x[d], T1 = tg1[d] - x[d];  // s - x == s + (-x) bit for bit
            const int K0 = pk0[d] + k[d], K1 = pk1[d] - k[d];
            const bool pos0 = 0.0 < T0, pos1 = 0.0 < T1;
            const int m0 = pos0 ? K0 : 0, m1 = pos1 ? K1 : 0;
            // get_used_columns (:847-869): some cluster is informative, and the positives sit where the gain is
            const bool used = (m0 > m1 ? m0 : m1) > 0xffff && 3 * ((m0 + m1) & 0xffff) > tp2[d];
            S += (used && pos0) ? T0 : 0.0;
            t1[d] = (used && pos1) ? T1 : 0.0;
            pert = pert || (T0 - x[d] != tg0[d]) || (T1 + x[d] != tg1[d]);
        }
#pragma unroll
        for (int d = 0; d < DMAX; d++) S += t1[d];
        S_out = S;
        pert_out = pert;
    };
    // get_lk of the start state: the same sum with nothing flipped
    double lk;
    {
        double t0[DMAX], t1[DMAX];
#pragma unroll
        for (int d = 0; d < DMAX; d++) {
            const bool pos0 = 0.0 < tg0[d], pos1 = 0.0 < tg1[d];
            const int in_use = ((pos0 ? pk0[d] : 0) + (pos1 ? pk1[d] : 0)) & 0xffff;
            const bool any = (pos0 && pk0[d] > 0xffff) || (pos1 && pk1[d] > 0xffff);
            const bool used = any && 3 * in_use > tp2[d];
            t0[d] = (used && pos0) ? tg0[d] : 0.0;
            t1[d] = (used && pos1) ? tg1[d] : 0.0;
        }
        double S = pair_at(c0);
#pragma unroll
        for (int d = 0; d < DMAX; d++) S += t0[d];
#pragma unroll
        for (int d = 0; d < DMAX; d++) S += t1[d];
        lk = unif64(S);
    }
    pair_up = pair_at(c0 + 1);
    pair_dn = pair_at(c0 > 0 ? c0 - 1 : 0);
    // ---- per read, for the current state: prop_l = get_lk with the read flipped; its entry in LDS: diff = prop_l - lk (the
    //      quantity `0f64 < diff || rng.gen_bool(diff.exp())` (:736) decides on, as f32), the hop-word bits that hold if the draw is
    //      above exp(diff) (.z), and those that hold anyway (.w: gen_bool(1.0) draws nothing, and exp(diff) == 1.0 exactly
    //      when diff >= -2^-54; the residue flag)
    double prop_l[NR];
    auto evaluate = [&]() {
#pragma unroll
        for (int r = 0; r < NR; r++) {
            bool pert;
            flipped_lk(sx[r], spk[r], __builtin_amdgcn_inverse_ballot_w64(lab[r]), prop_l[r], pert);
            const double diff = prop_l[r] - lk;
            u32x4_t e;
            e.x = __float_as_uint((float)diff);  // (rounded: within 3e-6 of diff wherever a threshold can lie -- the guard bands are 2e-3)
            e.y = 0;
            e.z = pert ? HW_REJ : HW_REJ | HW_SKIP;
            e.w = (diff >= -0x1p-54 ? HW_NODRAW : 0u) | (pert ? HW_PERT : 0u);
            if (lane + 64 * r < n) *(lds_vu32x4 *)(uintptr_t)(tab_lds + ((lane + 64 * r) << 4)) = e;
        }
    };
    // ---- the window.  Round 6: a window is a BLOCK of 64 consecutive stream positions, and the next block follows at + 64 whatever
    //      the proposals do (rounds 1-5: the next window started where the first proposal that did not end inside the window
    //      began, so nothing of it could be fetched before the walk had got there).  A proposal that starts in a block and ends in
    //      the next one is described by its own record like any other; its hop word carries HW_CROSS and, where it could have
    //      been stepped over, HW_SKIPX instead of HW_SKIP: the walk stops at it, counts it and goes on in the next block.  A
    //      block's look-up of its reads' entries is in flight while its two logarithms are computed, every position of the
    //      stream belongs to exactly one block (a window used to re-read the tail of its predecessor), and the move itself is
    //      ~60 instructions.  Solo chains -7 .. -10 % (profiles/r06_chain_solo.txt); requesting the next block's records a block
    //      ahead (JTK_K2_LOOKAHEAD 128) adds nothing: the chain is now bound by its producer wave.
    uint32_t w_base = 0;          // stream position of lane 0
    uint32_t w_idx = 0, w_w0 = 0; // per lane: the read the proposal starting here picks; its end (mod 64) | HW_CROSS, or HW_OUT
    float w_lrej = 0.0f, w_lacc = 0.0f;  // per lane: diff below w_lrej: certainly rejected; above w_lacc: certainly accepted
    uint32_t hopw = 0;
    uint32_t r_next = 0;          // per lane: the record of position w_base + 64 + lane
    uint32_t n_blocks = 0;        // (statistics build: windows loaded)
    auto hop_finish = [&](const u32x4_t tv) {
        const float diff = __uint_as_float(tv.x);
        // (tv.y is always 0 (evaluate).  It is OR-ed in so that all four registers of the 16-byte load stay live until the entry is
        // used: the compiler otherwise hands the dead one to the arithmetic that follows the load's issue, and the hardware then
        // has to wait for the load before that arithmetic may start)
        const uint32_t h = w_w0 | tv.w | tv.y | (diff < w_lrej ? tv.z : 0u) | (diff > w_lacc ? HW_ACC : 0u);
        hopw = (h & HW_CROSS) ? ((h & ~HW_SKIP) | ((h & HW_SKIP) << 7)) : h;
    };
    auto hop_words = [&]() { hop_finish(*(lds_cvu32x4 *)(uintptr_t)(tab_lds + (w_idx << 4))); };
    // the block whose records are `r`: per-lane registers and hop words (the entries of the block's reads are requested first,
    // the thresholds are computed while they are on their way)
    auto block_setup = [&](const uint32_t r) {
        w_idx = r & 127u;
        const u32x4_t tv = *(lds_cvu32x4 *)(uintptr_t)(tab_lds + (w_idx << 4));
        const uint32_t len = (r >> 7) & 63u, nxt = lane + len;
        const bool parsed = len != 0;
        w_w0 = parsed ? ((nxt & 63u) | (nxt >= 64u ? HW_CROSS : 0u)) : HW_OUT;
        // The 19 known bits u of the Bernoulli draw (its true value / 2^64 lies in [u, u + 2^-19)) against exp(diff), in the
        // log domain, with guard bands far wider than the errors of the hardware logarithm (v_log_f32: 1 ulp of a number below
        // 100, then one multiplication: < 2e-5) and of diff's rounding to f32 (< 3e-6 where a threshold can lie):
        //   diff < ln(u - 1.3e-6) - 2e-3  =>  exp(diff) * 1.002 < u - 1.3e-6: the draw is above p: rejected;  below -44.39
        //                                     exp(diff) * 2^64 < 1 (2^64 = e^44.3614), p_int == 0: rejected whatever the draw
        //   diff > ln(u + 2^-19 + 3e-7) + 2e-3  =>  exp(diff) > 1.002 (u + 2^-19 + 3e-7): the draw is below p: accepted
        const float u = (float)(r >> 13) * 0x1p-19f;
        const float lr = __builtin_amdgcn_logf(fmaxf(u - 1.3e-6f, 1e-30f)) * 0.6931472f - 2e-3f;  // (operands are normal numbers)
        const float la = __builtin_amdgcn_logf(u + (0x1p-19f + 3e-7f)) * 0.6931472f + 2e-3f;
        w_lrej = parsed ? fmaxf(lr, -44.39f) : -__builtin_inff();
        w_lacc = parsed ? la : __builtin_inff();
        hop_finish(tv);
    };
    // the records of [base, base + 128) exist and may not be overwritten
#ifndef JTK_K2_LOOKAHEAD
// Records that must exist beyond a block's base.  64: the block's own records are read when it is entered (default).  128: the
// next block's records are requested a block ahead (r_next) -- measured (profiles/r06_chain_solo.txt): nothing for the light
// kernel (the chain is bound by its producer wave, not by this round trip) and -9 % for the general kernel, whose ring holds
// 1,024 positions: the consumer then waits for the producer 64 positions earlier in every superblock.
#define JTK_K2_LOOKAHEAD 64u
#endif
    auto block_claim = [&](uint32_t base) {
        rng.pos = base;
        rng_release(rng, lane);
        rng_wait_rec(rng, base + JTK_K2_LOOKAHEAD);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        w_base = base;
    };
    auto block_first = [&](uint32_t base) {   // the chain's first block, or the one behind a proposal with scalar draws that went far
        block_claim(base);
        const uint32_t r = *(lds_vu32 *)(uintptr_t)(rec_lds + (((base + lane) & rn_mask) << 2));
#if JTK_K2_LOOKAHEAD >= 128u
        r_next = *(lds_vu32 *)(uintptr_t)(rec_lds + (((base + 64u + lane) & rn_mask) << 2));
#endif
        block_setup(r);
        n_blocks++;
    };
    auto block_advance = [&]() {              // on to the block at w_base + 64: its records came in while this one was walked
        block_claim(w_base + 64u);
#if JTK_K2_LOOKAHEAD >= 128u
        const uint32_t r = r_next;
        r_next = *(lds_vu32 *)(uintptr_t)(rec_lds + (((w_base + 64u + lane) & rn_mask) << 2));
#else   // (measurement: no request ahead -- the block's own records are read now)
        const uint32_t r = *(lds_vu32 *)(uintptr_t)(rec_lds + (((w_base + lane) & rn_mask) << 2));
#endif
        block_setup(r);
        n_blocks++;
    };
    double max = lk;
    unsigned long long argmax[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) argmax[r] = lab[r];
    evaluate();
    const uint32_t total = 2000u * n;
    uint32_t t = 0, p = 0;
    uint32_t n_events = 0;  // (reported per chunk: jtk_lc_debug_chain_profile)
    ST_T0();
    block_first(rng.pos);
    for (;;) {
        // ---- the walk, across blocks, up to the next proposal that cannot be skipped.  An inner loop of its own: it
        //      writes the window's registers and nothing of the chain's state, so the event below meets the back edge
        //      without a block of register moves between them.
        uint32_t hv = 0;
        bool done = false;
        for (;;) {
            if (t >= total) {
                done = true;
                break;
            }
            t += walk_rejected(hopw, p, total - t, hv);
            if (t >= total) {
                done = true;
                break;
            }
            if (!(hv & HW_SKIPX)) break;
            t++;             // certainly rejected, nothing left behind, ends in the next block: counted, and the walk goes on there
            p = hv & 63u;
            block_advance();
        }
        if (done) break;
#ifdef JTK_MCMC_STATS
        const unsigned long long g0c = __builtin_readcyclecounter();
#endif
        ST_MARK0();
        uint32_t e_idx, pos_v;  // the read it picks; stream position of the draw a Bernoulli test would compare
        if (!(hv & HW_OUT)) {
            e_idx = uni((uint32_t)__builtin_amdgcn_readlane((int)w_idx, (int)p));
            pos_v = w_base + (hv & 63u) + ((hv & HW_CROSS) ? 64u : 0u) - 1;
        } else {
            // the producer could not parse this one (it needs more look-ahead than it has, p ~ 2^-14): scalar draws and the
            // exact Bernoulli test
            scalar_proposal(rng, w_base + p, n, e_idx, pos_v);
            const u32x4_t tv = *(lds_cvu32x4 *)(uintptr_t)(tab_lds + (e_idx << 4));
            hv = HW_OUT | uni(tv.w);
        }
        // the read's lane: the likelihood of the flipped state and the read's signed row
        double proposed, x0[DMAX];
        int k0[DMAX];
        if (NR == 2 && e_idx >= 64) {
            proposed = readlane_f64(prop_l[NR - 1], e_idx & 63u);
#pragma unroll
            for (int d = 0; d < DMAX; d++) {
                x0[d] = readlane_f64(sx[NR - 1][d], e_idx & 63u);
                k0[d] = __builtin_amdgcn_readlane(spk[NR - 1][d], (int)(e_idx & 63u));
            }
        } else {
            proposed = readlane_f64(prop_l[0], e_idx & 63u);
#pragma unroll
            for (int d = 0; d < DMAX; d++) {
                x0[d] = readlane_f64(sx[0][d], e_idx & 63u);
                k0[d] = __builtin_amdgcn_readlane(spk[0][d], (int)(e_idx & 63u));
            }
        }
        ST_MARK(10);
        n_events++;
        uint32_t accept = (hv & (HW_NODRAW | HW_ACC)) ? 1u : 0u;
        if (!(hv & (HW_NODRAW | HW_ACC | HW_REJ))) {  // inside the guard bands (or a start without a record): the exact test
            rng_wait(rng, pos_v + 1);
            accept = ubool(bernoulli_exact(uni64(lds_ld64(&rng.ring[ring_slot(pos_v, rng.seg_log)])), proposed - lk)) ? 1u : 0u;  // (an out-of-line call returns in a vector register)
            ST_CNT(15, 1);
        }
        ST_MARK(11);
        uint32_t moved = 0;
        if (accept) {
#pragma unroll
            for (int d = 0; d < DMAX; d++) {
                tg0[d] = tg0[d] + x0[d];
                tg1[d] = tg1[d] - x0[d];
                pk0[d] += k0[d];
                pk1[d] -= k0[d];
            }
            const unsigned long long bit = 1ull << (e_idx & 63u);
            if (NR == 2 && e_idx >= 64) {
                c0 = (lab[NR - 1] & bit) ? c0 + 1 : c0 - 1;
                lab[NR - 1] ^= bit;
            } else {
                c0 = (lab[0] & bit) ? c0 + 1 : c0 - 1;  // the read sat in cluster 1: cluster 0 grows
                lab[0] ^= bit;
            }
            pair_up = pair_at(c0 + 1);
            pair_dn = pair_at(c0 > 0 ? c0 - 1 : 0);
            lk = proposed;
            if (ubool(max < lk)) {
                max = proposed;
#pragma unroll
                for (int r = 0; r < NR; r++) argmax[r] = lab[r];
            }
#pragma unroll
            for (int r = 0; r < NR; r++) {  // the read now flips the other way
                const bool mine = lane + 64 * r == e_idx;
#pragma unroll
                for (int d = 0; d < DMAX; d++) {
                    sx[r][d] = mine ? -sx[r][d] : sx[r][d];
                    spk[r][d] = mine ? -spk[r][d] : spk[r][d];
                }
            }
            moved = 1;
        } else if (hv & HW_PERT) {
            // flip back (:746) keeps the rounding residue: the sums move although nothing was accepted
#pragma unroll
            for (int d = 0; d < DMAX; d++) {
                tg0[d] = (tg0[d] + x0[d]) - x0[d];
                tg1[d] = (tg1[d] - x0[d]) + x0[d];
            }
            moved = 1;
        }
        t++;
        const uint32_t pos_next = pos_v + 1 - ((hv / HW_NODRAW) & 1u);
        ST_CNT(7, 1);
        ST_CNT(8, accept);
        ST_CNT(9, moved);
        ST_MARK(12);
        if (moved) evaluate();
        ST_MARK(13);
        const uint32_t off_next = pos_next - w_base;
        if (off_next >= 128u) {        // (only behind scalar draws that went on for more than a block)
            block_first(pos_next);
            p = 0;
        } else if (off_next >= 64u) {  // the entries are up to date (evaluate above): the new block's hop words are made from them
            p = off_next - 64u;
            block_advance();
        } else {
            p = off_next;
            if (moved) hop_words();
        }
        ST_MARK(14);
#ifdef JTK_MCMC_STATS
        ST_CNT(4, __builtin_readcyclecounter() - g0c);
#endif
    }
    ST_CNT(5, total);
    ST_CNT(6, n_blocks);
    ST_ADD(0);
    if (lane == 0) m.k2_stats[16] += n_events;
    rng.pos = w_base + p;
    rng_release(rng, lane);
#pragma unroll
    for (int r = 0; r < NR; r++)
        if (lane + 64 * r < n) m.assign[lane + 64 * r] = (uint8_t)((argmax[r] >> lane) & 1ull);
    wsync();
    *rng_io = rng;
    return max;
}

// get_lk (:785-795) of the labels in `assign` from freshly filled counters (reads in order, :752-758): what the
// reference compares the tracked maximum with before returning it (`assert!((max - lk).abs() < 0.0001)`, :759-760).
template <int K, bool HUGE>
__device__ __noinline__ double fresh_lk(LdsShape shape, uint32_t n_in, uint32_t D_in, double cov_in, uint32_t lane) {
    const uint32_t n = uni(n_in), D = uni(D_in);
    const double cov = unif64(cov_in);
    const Lds m = lds_carve<HUGE>(shape);
    const uint8_t *assign = m.assign;
    Counts<K> q;
    int clusters[K];
    fill_counts<K>(m, n, D, assign, q, clusters, lane);
    const unsigned long long usedm = __ballot(lane < D && column_used<K>(q));
    double S = 0.0;
#pragma unroll
    for (int c = 0; c < K; c++) {
        const uint32_t x = uni((uint32_t)clusters[c]);
        double mx = -__builtin_inf();
        for (int cc = 1; cc <= K; cc++) {
            const double lam = cov * (double)cc;
            mx = jtk_fmax(mx, (double)x * jtk_log(lam) - lam - m.lfact[x]);
        }
        S += unif64(mx);
    }
#pragma unroll
    for (int c = 0; c < K; c++) {
        unsigned long long mm = usedm;
        while (mm) {
            const uint32_t d = (uint32_t)__builtin_ctzll(mm);
            mm &= mm - 1;
            S += jtk_fmax(readlane_f64(q.tg[c], d), 0.0);
        }
    }
    return S;
}

template <int K, bool LIGHT, bool HUGE>
__device__ __forceinline__ double mcmc_chain_dispatch(const Lds &m, uint32_t n, uint32_t D, double cov, Rng &rng, uint32_t lane);

// mcmc_with_filter (:704-762): the chain, then the reference's closing self-check.  NaN = the reference panics.
template <int K, bool LIGHT, bool HUGE>
__device__ __forceinline__ double mcmc_with_filter(const Lds &m, uint32_t n, uint32_t D, double cov, Rng &rng, uint32_t lane) {
    const double max = mcmc_chain_dispatch<K, LIGHT, HUGE>(m, n, D, cov, rng, lane);
    const double fresh = fresh_lk<K, HUGE>(m.shape, n, D, cov, lane);
    if (!ubool(fabs(max - fresh) < 0.0001)) return __builtin_nan("");
    return max;
}

template <int K, bool LIGHT, bool HUGE>
__device__ __forceinline__ double mcmc_chain_dispatch(const Lds &m, uint32_t n, uint32_t D, double cov, Rng &rng, uint32_t lane) {
    if (HUGE) return mcmc_chain<K, false>(m, n, D, cov, rng, lane);  // the work area is in global memory (mcmc_kernel_huge)
    if (LIGHT) {
        // the light kernel (mcmc_kernel_light) holds only the chain variants that fit 168 registers; chain_split_kernel
        // sends it nothing else.  NaN = the chunk fails, loudly, should that ever not hold.
        if (!(K == 2 && n <= JTK_LIGHT_MAX_READS && D >= 1 && D <= JTK_LIGHT_MAX_DIM)) return __builtin_nan("");
        rng_set_parse_mode(rng, PM_K2, lane);
        const K2Mem km = {m.data, m.lfact, m.assign, m.fbuf, m.k2_stats};
        if (n <= 63) {
            if (D == 1) return mcmc_chain_k2<1, 1>(km, n, D, cov, &rng, lane);
            return mcmc_chain_k2<2, 1>(km, n, D, cov, &rng, lane);
        }
        if (D == 1) return mcmc_chain_k2<1, 2>(km, n, D, cov, &rng, lane);  // 64 .. 127 reads: two table registers
        return mcmc_chain_k2<2, 2>(km, n, D, cov, &rng, lane);
    }
    if (K == 2 && n <= 127 && D >= 1 && D <= 8) {
        rng_set_parse_mode(rng, PM_K2, lane);
        const K2Mem km = {m.data, m.lfact, m.assign, m.fbuf, m.k2_stats};
        if (n <= 63) {
            if (D == 1) return mcmc_chain_k2<1, 1>(km, n, D, cov, &rng, lane);
            if (D == 2) return mcmc_chain_k2<2, 1>(km, n, D, cov, &rng, lane);
            if (D == 3) return mcmc_chain_k2<3, 1>(km, n, D, cov, &rng, lane);
            if (D == 4) return mcmc_chain_k2<4, 1>(km, n, D, cov, &rng, lane);
            return mcmc_chain_k2<8, 1>(km, n, D, cov, &rng, lane);
        }
        if (D <= 2) return mcmc_chain_k2<2, 2>(km, n, D, cov, &rng, lane);
        if (D <= 4) return mcmc_chain_k2<4, 2>(km, n, D, cov, &rng, lane);
        return mcmc_chain_k2<8, 2>(km, n, D, cov, &rng, lane);
    }
    rng_set_parse_mode(rng, (uint32_t)K, lane);
    return mcmc_chain_tab<K>(m.shape, n, D, cov, &rng, lane);
}

// get_read_lk_gains (:381-408): used columns -> used[], per-read gain -> fbuf[]
template <int K>
__device__ __forceinline__ void get_read_lk_gains(const Lds &m, uint32_t n, uint32_t D, const uint8_t *assign, uint8_t *used,
                                  uint32_t lane) {
    Counts<K> q;
    int clusters[K];
    fill_counts<K>(m, n, D, assign, q, clusters, lane);
    const bool u = lane < D && column_used<K>(q);
    if (lane < D) used[lane] = u ? 1 : 0;
#pragma unroll
    for (int c = 0; c < K; c++)
        if (lane < D) m.val[c * D + lane] = (u && JTK_POS_THR < q.tg[c]) ? 1.0 : 0.0;  // column counts for cluster c
    wsync();
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t a = assign[i];
        double s = 0.0;
        for (uint32_t d = 0; d < D; d++)
            if (m.val[a * D + d] != 0.0) s += m.data[i * D + d];
        m.fbuf[i] = s;
    }
    wsync();
}

// get_likelihood_gain (:353-379): out[i*K + c]
template <int K>
__device__ __forceinline__ void get_likelihood_gain(const Lds &m, uint32_t n, uint32_t D, const uint8_t *assign, double *out,
                                    uint32_t lane) {
    Counts<K> q;
    int clusters[K];
    fill_counts<K>(m, n, D, assign, q, clusters, lane);
    const bool u = lane < D && column_used<K>(q);
#pragma unroll
    for (int c = 0; c < K; c++)
        if (lane < D) m.val[c * D + lane] = (u && JTK_POS_THR < q.tg[c]) ? 1.0 : 0.0;
    wsync();
    for (uint32_t i = lane; i < n; i += 64)
        for (int c = 0; c < K; c++) {
            double s = 0.0;
            for (uint32_t d = 0; d < D; d++)
                if (m.val[c * D + d] != 0.0) s += m.data[i * D + d];
            out[i * K + c] = s;
        }
    wsync();
}

// mcmc_clustering (:649-670): labels -> m.best, per-read gains -> m.fbuf, used columns -> m.used
template <int K, bool LIGHT, bool HUGE>
__device__ __forceinline__ bool mcmc_clustering(const Lds &m, uint32_t n, uint32_t D, double cov, Rng &rng, double *score,
                                uint32_t lane) {
    double best = 0.0;
    bool have = false;
    for (int it = 0; it < 20; it++) {
        if (!kmeans(m, n, D, K, rng, lane)) return false;
        const double lk = mcmc_with_filter<K, LIGHT, HUGE>(m, n, D, cov, rng, lane);
        if (ubool(lk != lk)) return false;  // the reference panicked inside mcmc_with_filter
#ifdef JTK_DEBUG_LK
        if (lane == 0 && n == 65) printf("DEVLK n %u D %u it %d lk %.17g pos %u\n", n, D, it, lk, rng.pos);
#endif
        if (!have || !(lk < best)) {  // max_by: the last maximum wins
            best = lk;
            have = true;
            for (uint32_t i = lane; i < n; i += 64) m.best[i] = m.assign[i];
            wsync();
        }
    }
    get_read_lk_gains<K>(m, n, D, m.best, m.used, lane);
    // cluster_lk = sum_c max_poisson_lk(count_c, cov, 1, K)
    double cluster_lk = 0.0;
    for (int c = 0; c < K; c++) {
        uint32_t cnt = 0;
        for (uint32_t i = 0; i < n; i++) cnt += m.best[i] == c ? 1u : 0u;
        double mx = -__builtin_inf();
        for (int cc = 1; cc <= K; cc++) {
            const double lam = cov * (double)cc;
            mx = jtk_fmax(mx, (double)cnt * jtk_log(lam) - lam - m.lfact[cnt]);
        }
        cluster_lk += mx;
    }
    *score = best - cluster_lk;
    return true;
}

// out of line: one candidate cluster count per call keeps the kernel body (k-means, model selection, posteriors) small
template <int K, bool LIGHT, bool HUGE = false>
__device__ __attribute__((noinline)) bool run_k(LdsShape shape, uint32_t n_in, uint32_t D_in, double cov_in, Rng &rng, double *score,
                                                uint32_t lane) {
    const uint32_t n = uni(n_in), D = uni(D_in);
    const double cov = unif64(cov_in);
    const Lds m = lds_carve<HUGE>(shape);
    return mcmc_clustering<K, LIGHT, HUGE>(m, n, D, cov, rng, score, lane);
}

template <bool LIGHT, bool HUGE>
__device__ __forceinline__ bool run_k_dyn(uint32_t k, LdsShape m, uint32_t n, uint32_t D, double cov, Rng &rng, double *score,
                          uint32_t lane) {
    if (LIGHT) return k == 2 ? run_k<2, true>(m, n, D, cov, rng, score, lane) : false;
    switch (k) {
        case 2: return run_k<2, false, HUGE>(m, n, D, cov, rng, score, lane);
        case 3: return run_k<3, false, HUGE>(m, n, D, cov, rng, score, lane);
        case 4: return run_k<4, false, HUGE>(m, n, D, cov, rng, score, lane);
        case 5: return run_k<5, false, HUGE>(m, n, D, cov, rng, score, lane);
        case 6: return run_k<6, false, HUGE>(m, n, D, cov, rng, score, lane);
        case 7: return run_k<7, false, HUGE>(m, n, D, cov, rng, score, lane);
        default: return false;
    }
}

template <bool LIGHT>
__device__ __forceinline__ void likelihood_gain_dyn(uint32_t k, const Lds &m, uint32_t n, uint32_t D, const uint8_t *assign,
                                    double *out, uint32_t lane) {
    if (LIGHT) {  // copy number 2: one or two clusters
        if (k == 1) get_likelihood_gain<1>(m, n, D, assign, out, lane);
        else get_likelihood_gain<2>(m, n, D, assign, out, lane);
        return;
    }
    switch (k) {
        case 1: get_likelihood_gain<1>(m, n, D, assign, out, lane); break;
        case 2: get_likelihood_gain<2>(m, n, D, assign, out, lane); break;
        case 3: get_likelihood_gain<3>(m, n, D, assign, out, lane); break;
        case 4: get_likelihood_gain<4>(m, n, D, assign, out, lane); break;
        case 5: get_likelihood_gain<5>(m, n, D, assign, out, lane); break;
        case 6: get_likelihood_gain<6>(m, n, D, assign, out, lane); break;
        default: get_likelihood_gain<7>(m, n, D, assign, out, lane); break;
    }
}

// (trace instantiation only) get_read_lk_gains for a run-time cluster count 2 .. 7
__device__ __forceinline__ void read_lk_gains_dyn(uint32_t k, const Lds &m, uint32_t n, uint32_t D, const uint8_t *assign, uint8_t *used,
                                                  uint32_t lane) {
    switch (k) {
        case 2: get_read_lk_gains<2>(m, n, D, assign, used, lane); break;
        case 3: get_read_lk_gains<3>(m, n, D, assign, used, lane); break;
        case 4: get_read_lk_gains<4>(m, n, D, assign, used, lane); break;
        case 5: get_read_lk_gains<5>(m, n, D, assign, used, lane); break;
        case 6: get_read_lk_gains<6>(m, n, D, assign, used, lane); break;
        default: get_read_lk_gains<7>(m, n, D, assign, used, lane); break;
    }
}

__device__ __forceinline__ double gains_expected(const jtk_gains_t *g, uint32_t homop_len, int dt) {
    if (homop_len == 0) homop_len = 1;
    const uint32_t h = homop_len < g->max_homopolymer_len ? homop_len : g->max_homopolymer_len;
    return dt == JTK_DIFF_SUBST ? g->subst[h - 1].gain
                                : (dt == JTK_DIFF_DEL ? g->deletions[h - 1].gain : g->insertions[h - 1].gain);
}

// one workgroup of two waves per chunk: wave 0 runs the algorithm, wave 1 feeds it proposals
// Register budget: two waves per SIMD (<= 256 registers; the kernel needs 248).  A chain workgroup is two latency-bound waves
// that leave their SIMDs idle most of the time: at 360 registers (the legacy chain inlined) a chain wave had its SIMD to itself
// and 625 workgroups shut every other kernel out of the machine; at 248 two of them share a SIMD, or one sits beside a
// pair-HMM wave of another batch (152 registers) -- bench.py overlaps batches: 1,680 -> 1,820 chunks/s.
//
// Two entry points share the body.  `mcmc_kernel` holds every chain variant (248 registers).  `mcmc_kernel_light` holds only
// what a diploid pile-up of <= 63 reads with one or two variant columns needs -- 80 % of the headline's chunks have D <= 1,
// 97 % D <= 2 -- and fits 168 registers: three of its waves share a SIMD, or one of them sits beside TWO pair-HMM waves of
// another batch (168 + 2 x 168 <= 512) where a 248-register chain wave leaves room for one.  Which chunk goes where is
// decided on the device (chain_split_kernel: D is known only after the filter), with no host round trip.
#ifndef JTK_MCMC_WAVES
#define JTK_MCMC_WAVES 2
#endif
// TRACE = true (mcmc_kernel_trace, jtk_lc_session_trace): the same clustering once more for ONE chunk, leaving what the reference's
// trace! rows of cluster_filtered_variants need in `trace` (doubles): [0] = first k, [1] = last k of the RANGE row (:236), [2] = number
// of records; record r at [8 + 16 r]: k, score, expected_gain, improved_reads (the LK rows :250,:256), accepted?, then the k cluster
// sizes (COUNTS :262); from [JTK_TRACE_OLD] the n per-read gains of the accepted clustering (read_lk_gains :229, which feeds only
// improved_reads).  The product instantiations compile none of it.
#define JTK_TRACE_REC 8
#define JTK_TRACE_REC_LEN 16
#define JTK_TRACE_OLD (JTK_TRACE_REC + 8 * JTK_TRACE_REC_LEN)
template <bool LIGHT, bool HUGE = false, bool TRACE = false>
__device__ __forceinline__ void mcmc_body(const ChunkMeta *chunks, ChunkState *state,
                                                  const jtk_lc_params_t *params, const double *feat_all,
                                                  const uint32_t *vtype_all, const uint64_t *vt_off_all,
                                                  uint32_t vt_stride_mode, uint32_t *label_all, double *post_all,
                                                  uint32_t post_stride, double *lg_all, const uint64_t *lg_off,
                                                  uint32_t lds_n, uint32_t lds_d, uint32_t lds_k, uint32_t seg_log_in,
                                                  uint32_t flags, const uint64_t *rng_resume, const uint32_t *order,
                                                  const uint32_t *order_count, unsigned char *ws_base = nullptr,
                                                  const uint64_t *ws_off = nullptr, double *trace = nullptr) {
    // workgroups are dispatched in blockIdx order: `order` lists the chunks with the longest chains first (their
    // length is 20 x 2000 x n proposals per candidate k), so that on ragged batches the kernel does not end on a
    // long chain that started late.  `order_count`, if given, is the device-side length of the list (the grid is the
    // upper bound the host knows).
    if (order_count && blockIdx.x >= uni(*order_count)) return;
    const uint32_t seg_log = uni(seg_log_in);  // the ring's geometry: JTK_SEG_LOG_LIGHT / _GENERAL
    const uint32_t ci = order ? order[blockIdx.x] : blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ChunkState *st = &state[ci];
    if (st->status != 0) return;
    const ChunkMeta cm = chunks[ci];
    // everything that steers control flow is made provably wave-uniform (scalar registers, scalar branches)
    const uint32_t n = uni(cm.n_reads), D = uni(st->dim), copy_num = uni(cm.copy_num);
    const double coverage = unif64(params->haploid_coverage);
    uint32_t *label = label_all + cm.read_first;
    double *post = post_all + (uint64_t)cm.read_first * post_stride;
    // ---- trivial outcomes (pseudo_mcmc.rs:86-88, :221-225)
    if (copy_num < 2 || D == 0 || n <= copy_num) {
        if (wave != 0) return;
        for (uint32_t i = lane; i < n; i += 64) {
            label[i] = 0;
            for (uint32_t c = 0; c < post_stride; c++) post[(uint64_t)i * post_stride + c] = 0.0;
        }
        if (lane == 0) {
            st->score = 0.0;
            st->k = 1;
            st->draws = 0;
        }
        return;
    }
    if (copy_num > JTK_MAX_COPY || copy_num > lds_k || (!HUGE && n > JTK_MAX_PILEUP) || n > lds_n || D > lds_d) {
        if (threadIdx.x == 0) st->status = JTK_ERR_UNSUPPORTED;
        return;
    }
    // ---- LDS carve (again after every out-of-line call: the pointers are cheaper to re-make than to keep alive across it)
    (void)flags;
    // (mcmc_kernel_huge: lds_n / lds_d / lds_k are this chunk's own sizes and size its slice of the global workspace)
    const LdsShape shape = {HUGE ? n : lds_n, HUGE ? D : lds_d, HUGE ? (copy_num < 2 ? 2u : copy_num) : lds_k, seg_log,
                            HUGE ? (uint64_t)(uintptr_t)(ws_base + ws_off[blockIdx.x]) : 0ull};
    Lds m = lds_carve<HUGE>(shape);
    if (threadIdx.x == 0) {
        lds_st32(&m.ctl->rd, 0);
        lds_st32(&m.ctl->quit, 0);
        lds_st32(&m.ctl->wr, 0);
        lds_st32(&m.ctl->wp, 0);
        lds_st32(&m.ctl->parse_n, n);
        lds_st32(&m.ctl->parse_from, 0);
        lds_st32(&m.ctl->pmode, 0);      // no records until a chain asks for them (rng_set_parse_mode)
        lds_st32(&m.ctl->wp_epoch, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
#ifdef JTK_MCMC_PRIO_PRODUCER  // experiments: wave priority of the two halves of a chain workgroup (s_setprio 0..3)
    if (wave == 1) __builtin_amdgcn_s_setprio(JTK_MCMC_PRIO_PRODUCER);
#endif
#ifdef JTK_MCMC_PRIO_CONSUMER
    if (wave == 0) __builtin_amdgcn_s_setprio(JTK_MCMC_PRIO_CONSUMER);
#endif
    if (wave == 1) {  // Xoshiro256StarStar::seed_from_u64(chunk.id * 3490)  (local_clustering/mod.rs:97)
        producer_main(m.ctl, m.ring, m.rec, seg_log, uni64(cm.chunk_id) * 3490ULL,
                      rng_resume ? rng_resume + 4 * (uint64_t)ci : nullptr, lane);
        return;
    }
    const double *feat = feat_all + cm.feat_off;
    // LKCount::{add,sub} assert `x.abs() < POS_THR` for a value that is neither above POS_THR nor below -POS_THR
    // (:830,:841: exactly +-POS_THR, or NaN), and mcmc_with_filter asserts that its size table has no NaN (:714-715:
    // x ln(lambda) - lambda - ln x! is NaN at x = 0 unless 0 < lambda < inf): the reference panics, the chunk fails.
    bool bad_value = false;
    for (uint32_t e = lane; e < n * D; e += 64) {
        const double x = feat[e];
        m.data[e] = x;
        bad_value = bad_value || (!(JTK_POS_THR < x) && !(x < -JTK_POS_THR) && !(fabs(x) < JTK_POS_THR));
    }
    if (ubool(__ballot(bad_value) != 0ull) || !(coverage > 0.0 && coverage < __builtin_inf())) {
        if (lane == 0) {
            lds_st32(&m.ctl->quit, 1);
            st->status = JTK_ERR_CHUNK_FAILED;
        }
        return;
    }
    // lfact[x] = sum_{c=1..x} ln c, summed left to right as poisson_lk does (:636-638)
    if (lane == 0) {
        double s = 0.0;
        m.lfact[0] = 0.0;
        for (uint32_t c = 1; c <= n; c++) {
            s += jtk_log((double)c);
            m.lfact[c] = s;
        }
    }
    for (uint32_t d = lane; d < D; d += 64) m.prev_used[d] = 0;
    for (uint32_t i = lane; i < n; i += 64) m.accepted[i] = 0;
    if (lane < K2_STAT_SLOTS) m.k2_stats[lane] = 0;
    const unsigned long long chain_t0 = __builtin_readcyclecounter();
    wsync();
    const uint32_t *vt = vtype_all + 2 * ((uint64_t)ci * JTK_MAX_DIM);
    if (vt_stride_mode) vt = vtype_all + 2 * vt_off_all[ci];
    // ---- per-chunk RNG (local_clustering/mod.rs:97): the consumer's position in the producer's stream
    Rng rng;
    rng.pos = 0;
    rng.wr_seen = 0;
    rng.wp_seen = 0;
    rng.pmode = 0;
    rng.win_base = 0xffffff00u;  // nothing held yet
    rng.seg_log = seg_log;
    rng.win = 0;
#ifdef JTK_MCMC_STATS
    rng.waits = 0;
#endif
    rng.ctl = m.ctl;
    rng.ring = m.ring;
    rng.rec = m.rec;
    // ---- cluster_filtered_variants (:213-274)
    const double per_cluster_cov = unif64(cm.local_coverage);
    double max = 0.0;
    uint32_t max_k = 1;
    const uint32_t end = copy_num < 1 + 2 * D ? copy_num : 1 + 2 * D;
    const uint32_t start = (end > 5 ? end : 5) - 3;
    bool failed = false;
    uint32_t trace_n = 0;  // (TRACE) records written
    if constexpr (TRACE) {
        if (lane == 0) {
            trace[0] = (double)start;
            trace[1] = (double)end;
            trace[2] = 0.0;
        }
        for (uint32_t i = lane; i < n; i += 64) trace[JTK_TRACE_OLD + i] = 0.0;
        wsync();
    }
    for (uint32_t k = start; k <= end; k++) {
        double score;
        const bool ran = run_k_dyn<LIGHT, HUGE>(k, shape, n, D, coverage, rng, &score, lane);
        m = lds_carve<HUGE>(shape);
        if (!ran) {
            failed = true;
            break;
        }
        // result of this k: labels m.best, gains m.fbuf (unused downstream), used columns m.used
        if (k == 2) {
            // use_highest_gain (:673-693)
            double gbest = 0.0;
            uint32_t max_idx = 0;
            for (uint32_t d = 0; d < D; d++) {
                double gsum = 0.0;
                for (uint32_t i = 0; i < n; i++) gsum += jtk_fmax(m.data[i * D + d], 0.0);
                if (d == 0 || !(gsum < gbest)) {
                    gbest = gsum;
                    max_idx = d;
                }
            }
            for (uint32_t i = lane; i < n; i += 64) m.tmp_asn[i] = 0.0 < m.data[i * D + max_idx] ? 1 : 0;
            wsync();
            // keep the mcmc result aside: fbuf is overwritten by get_read_lk_gains
            get_read_lk_gains<2>(m, n, D, m.tmp_asn, m.tmp_used, lane);
            double hscore = 0.0;
            for (uint32_t i = 0; i < n; i++) hscore += m.fbuf[i];
            if (score < hscore) {
                score = hscore;
                for (uint32_t i = lane; i < n; i += 64) m.best[i] = m.tmp_asn[i];
                for (uint32_t d = lane; d < D; d += 64) m.used[d] = m.tmp_used[d];
                wsync();
            }
        }
        // expected_gains (:286-306)
        bool no_new = true;
        for (uint32_t d = 0; d < D; d++) no_new = no_new && (m.prev_used[d] == m.used[d]);
        double expt = 0.0;
        for (uint32_t d = 0; d < D; d++) {
            const bool check = ((!m.prev_used[d]) && m.used[d]) || no_new;
            const double v = check ? gains_expected(&params->gains, vt[2 * d], (int)vt[2 * d + 1]) : 0.0000001;
            if (d == 0 || !(v < expt)) expt = v;
        }
        const double expected_gain = jtk_fmax(0.8 * expt, 0.1) * per_cluster_cov + 0.1;
        if constexpr (TRACE) {
            // new_lk_gains of the clustering this k ended with (at k == 2 fbuf may hold the other candidate's): get_read_lk_gains of
            // m.best once more, then min_gain (:276-284: first minimum, the value only) and count_improved_reads (:308-312)
            read_lk_gains_dyn(k, m, n, D, m.best, m.tmp_used, lane);
            double min_gain = 1.0;
            bool have_min = false;
            for (uint32_t d = 0; d < D; d++)
                if (m.used[d]) {
                    const double v = gains_expected(&params->gains, vt[2 * d], (int)vt[2 * d + 1]) / 3.0;
                    if (!have_min || v < min_gain) min_gain = v;
                    have_min = true;
                }
            uint32_t improved = 0;
            for (uint32_t i = 0; i < n; i++) improved += trace[JTK_TRACE_OLD + i] + min_gain < m.fbuf[i] ? 1u : 0u;
            const bool accept = expected_gain < score - max;
            double *rec = trace + JTK_TRACE_REC + (uint64_t)trace_n * JTK_TRACE_REC_LEN;
            if (lane == 0 && trace_n < 8) {
                rec[0] = (double)k;
                rec[1] = score;
                rec[2] = expected_gain;
                rec[3] = (double)improved;
                rec[4] = accept ? 1.0 : 0.0;
                for (uint32_t c = 0; c < k; c++) {
                    uint32_t cnt = 0;
                    for (uint32_t i = 0; i < n; i++) cnt += m.best[i] == c ? 1u : 0u;
                    rec[5 + c] = (double)cnt;
                }
                trace[2] = (double)(trace_n + 1);
            }
            trace_n++;
            if (accept)
                for (uint32_t i = lane; i < n; i += 64) trace[JTK_TRACE_OLD + i] = m.fbuf[i];
            wsync();
        }
        if (expected_gain < score - max) {
            wsync();
            for (uint32_t i = lane; i < n; i += 64) m.accepted[i] = m.best[i];
            for (uint32_t d = lane; d < D; d += 64) m.prev_used[d] = m.used[d];
            max = score;
            max_k = k;
            wsync();
        } else {
            break;
        }
    }
    // stop the producer wave
    if (lane == 0) {
        lds_st32(&m.ctl->quit, 1);
        st->draws = rng.pos;  // stream positions consumed: where the chunk's next clustering() call resumes
        st->chain_cycles = __builtin_readcyclecounter() - chain_t0;
        st->chain_events = (uint32_t)m.k2_stats[16];
    }
#ifdef JTK_MCMC_STATS
    if (lane == 0)
        printf("K2WAIT chunk %u waits %u\n", ci, rng.waits);
    if (lane == 0)
        printf("K2STAT chunk %u n %u D %u cyc %llu walk %llu win %llu event %llu rebuild %llu steps %llu windows %llu events %llu accepts %llu changed %llu head %llu bern %llu book %llu tables %llu hops %llu exact %llu\n",
               ci, n, D, m.k2_stats[0], m.k2_stats[1], m.k2_stats[2], m.k2_stats[3], m.k2_stats[4], m.k2_stats[5],
               m.k2_stats[6], m.k2_stats[7], m.k2_stats[8], m.k2_stats[9], m.k2_stats[10], m.k2_stats[11], m.k2_stats[12],
               m.k2_stats[13], m.k2_stats[14], m.k2_stats[15]);
#endif
    if (failed) {
        if (lane == 0) st->status = JTK_ERR_CHUNK_FAILED;
        return;
    }
    // ---- likelihood gains of the accepted clustering, re-assignment, posterior (:272, :98-105, :342-347)
    double *lg = lg_all + lg_off[ci];  // n x max_k
    likelihood_gain_dyn<LIGHT>(max_k, m, n, D, m.accepted, lg, lane);
    __threadfence_block();
    for (uint32_t i = lane; i < n; i += 64) {
        double *lks = lg + (uint64_t)i * max_k;
        uint32_t asn = m.accepted[i], bi = 0;
        for (uint32_t c = 1; c < max_k; c++)
            if (!(lks[c] < lks[bi])) bi = c;
        if (lks[asn] + 0.001 < lks[bi]) asn = bi;
        // logsumexp (misc.rs:84-92)
        double mx = lks[0];
        for (uint32_t c = 1; c < max_k; c++)
            if (!(lks[c] < mx)) mx = lks[c];
        double sum = 0.0;
        for (uint32_t c = 0; c < max_k; c++) sum += jtk_exp(lks[c] - mx);
        const double total = mx + jtk_log(sum);
        label[i] = asn;
        for (uint32_t c = 0; c < post_stride; c++)
            post[(uint64_t)i * post_stride + c] = c < max_k ? lks[c] - total : 0.0;
    }
    if (lane == 0) {
        st->score = max;
        st->k = max_k;
    }
}

#define MCMC_KERNEL_PARAMS                                                                                                  \
    const ChunkMeta *chunks, ChunkState *state, const jtk_lc_params_t *params, const double *feat_all,                      \
        const uint32_t *vtype_all, const uint64_t *vt_off_all, uint32_t vt_stride_mode, uint32_t *label_all,                \
        double *post_all, uint32_t post_stride, double *lg_all, const uint64_t *lg_off, uint32_t lds_n, uint32_t lds_d,     \
        uint32_t lds_k, uint32_t seg_log, uint32_t flags, const uint64_t *rng_resume, const uint32_t *order,                \
        const uint32_t *order_count
#define MCMC_KERNEL_ARGS                                                                                                    \
    chunks, state, params, feat_all, vtype_all, vt_off_all, vt_stride_mode, label_all, post_all, post_stride, lg_all,       \
        lg_off, lds_n, lds_d, lds_k, seg_log, flags, rng_resume, order, order_count
__global__ __launch_bounds__(128, JTK_MCMC_WAVES) void mcmc_kernel(MCMC_KERNEL_PARAMS) { mcmc_body<false>(MCMC_KERNEL_ARGS); }
// Pile-ups whose work area does not fit a CU's LDS, or of more than JTK_MAX_PILEUP reads: the per-read arrays live in a
// global-memory workspace (ws_base + ws_off[block]), the chain is the one-proposal-per-iteration one.  One wave per SIMD: the
// seven inlined instantiations of that chain need ~360 registers, and nothing here is built for speed.
__global__ __launch_bounds__(128, 1) void mcmc_kernel_huge(MCMC_KERNEL_PARAMS, unsigned char *ws_base, const uint64_t *ws_off) {
    mcmc_body<false, true>(MCMC_KERNEL_ARGS, ws_base, ws_off);
}
// jtk_lc_session_trace: ONE chunk through the global-workspace body (any pile-up size), with the trace records
__global__ __launch_bounds__(128, 1) void mcmc_kernel_trace(MCMC_KERNEL_PARAMS, unsigned char *ws_base, const uint64_t *ws_off,
                                                            double *trace) {
    mcmc_body<false, true, true>(MCMC_KERNEL_ARGS, ws_base, ws_off, trace);
}
#ifndef JTK_MCMC_LIGHT_WAVES
#ifdef JTK_MCMC_STATS
#define JTK_MCMC_LIGHT_WAVES 2  // the statistics build prints from the kernel body: it does not fit 168 registers
#else
#define JTK_MCMC_LIGHT_WAVES 3
#endif
#endif
__global__ __launch_bounds__(128, JTK_MCMC_LIGHT_WAVES) void mcmc_kernel_light(MCMC_KERNEL_PARAMS) { mcmc_body<true>(MCMC_KERNEL_ARGS); }

// One wave splits a launch's chunk list (order[] or 0 .. count-1) into the chunks the light kernel can run and the rest,
// keeping the order (longest chain first) in both: out = counts[2] | light[count] | heavy[count].  Chunks with a trivial
// outcome (no variant column, failed earlier, copy number < 2) go to the light list: they return at once in either kernel.
__global__ __launch_bounds__(64) void chain_split_kernel(uint32_t count, const uint32_t *order, const ChunkMeta *chunks,
                                                         const ChunkState *state, uint32_t *out) {
    const uint32_t lane = threadIdx.x;
    uint32_t *light = out + 2, *heavy = out + 2 + count;
    uint32_t nl = 0, nh = 0;
    for (uint32_t base = 0; base < count; base += 64) {
        const uint32_t idx = base + lane;
        const bool valid = idx < count;
        uint32_t ci = 0;
        bool is_light = false;
        if (valid) {
            ci = order ? order[idx] : idx;
            const uint32_t n = chunks[ci].n_reads, copy_num = chunks[ci].copy_num, D = state[ci].dim;
            const bool trivial = state[ci].status != 0 || copy_num < 2 || D == 0 || n <= copy_num;
            is_light = trivial || (copy_num == 2 && n <= JTK_LIGHT_MAX_READS && D <= JTK_LIGHT_MAX_DIM);
        }
        const uint64_t ml = __ballot(valid && is_light), mh = __ballot(valid && !is_light);
        const uint64_t below = (1ull << lane) - 1ull;
        if (valid) {
            if (is_light) light[nl + __popcll(ml & below)] = ci;
            else heavy[nh + __popcll(mh & below)] = ci;
        }
        nl += __popcll(ml);
        nh += __popcll(mh);
    }
    if (lane == 0) {
        out[0] = nl;
        out[1] = nh;
    }
}

}  // namespace

// LDS work area of one chunk for a ring of 2 x (64 << seg_log) positions (the producer's jump table lives in global memory / L2).
static size_t mcmc_lds_core(uint32_t lds_n, uint32_t lds_d, uint32_t lds_k, uint32_t seg_log = JTK_SEG_LOG_LIGHT) {
    auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t npad = (lds_n + 63u) & ~63u;
    const size_t RN = RN_OF(seg_log);
    return al(sizeof(RCtl)) + al(sizeof(uint64_t) * RN) + al(sizeof(uint32_t) * RN) +
           al(K2_STAT_SLOTS * 8) + al((size_t)lds_n * lds_d * 8) + 2 * al((size_t)(lds_n + 1) * 8) +
           2 * al((size_t)JTK_MAX_COPY * lds_d * 8) + 5 * al(lds_n) + 3 * al(lds_d) +   // (fbuf / cum live inside stab)
           al((size_t)lds_k * npad * 8) + al(npad * 4) + al((size_t)lds_k * sizeof(SzEnt)) + al((size_t)lds_d * lds_k * 16) +
           al((size_t)lds_d * 16);
}
static uint32_t clamp_k(uint32_t lds_k) { return lds_k < 2 ? 2 : (lds_k > JTK_MAX_COPY ? JTK_MAX_COPY : lds_k); }
// mcmc_kernel_huge: the LDS in front of the sized arrays (ring + control block) and the global workspace of one chunk (an upper
// bound: everything sized by n, d, k -- lds_carve keeps what fits JTK_HUGE_LDS in LDS)
static size_t mcmc_lds_fixed() {
    auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t RN = RN_OF(JTK_SEG_LOG_LIGHT);
    return al(sizeof(RCtl)) + al(sizeof(uint64_t) * RN) + al(sizeof(uint32_t) * RN) + al(K2_STAT_SLOTS * 8);
}
size_t mcmc_ws_bytes(uint32_t n, uint32_t d, uint32_t k) {
    k = clamp_k(k);
    return ((mcmc_lds_core(n, d, k) - mcmc_lds_fixed()) + 255) & ~(size_t)255;
}
// (the session sorts chunks into launch classes by this number, which assumes the light kernel's 24 KiB ring; the general kernel
// is launched with 12 KiB less (launch_mcmc), so the classing is conservative by that much and jtk_lc_timing_t.chain_lds_bytes
// reports an upper bound -- ADVICE round 5; left as it is: tests pin the class boundaries at these sizes)
size_t mcmc_lds_bytes(uint32_t lds_n, uint32_t lds_d, uint32_t lds_k) { return mcmc_lds_core(lds_n, lds_d, clamp_k(lds_k)); }

// ---- host: the byte-digit table of M^(63*SEG), from nothing but the generator's own step function
namespace {
struct V256 {
    uint64_t w[4];
};
V256 host_xo_step(V256 v) {
    uint64_t s0 = v.w[0], s1 = v.w[1], s2 = v.w[2], s3 = v.w[3];
    const uint64_t t = s1 << 17;
    s2 ^= s0;
    s3 ^= s1;
    s1 ^= s2;
    s0 ^= s3;
    s2 ^= t;
    s3 = (s3 << 45) | (s3 >> 19);
    return V256{{s0, s1, s2, s3}};
}
struct M256 {
    V256 col[256];  // image of unit vector b (bit b & 63 of word b >> 6)
};
V256 m_apply(const M256 &a, const V256 &v) {
    V256 r{{0, 0, 0, 0}};
    for (int b = 0; b < 256; b++)
        if ((v.w[b >> 6] >> (b & 63)) & 1ull)
            for (int q = 0; q < 4; q++) r.w[q] ^= a.col[b].w[q];
    return r;
}
void m_mul(const M256 &a, const M256 &b, M256 &out) {  // out = a * b
    for (int i = 0; i < 256; i++) out.col[i] = m_apply(a, b.col[i]);
}
std::vector<uint64_t> build_jump_table(uint32_t SEG) {
    std::vector<uint64_t> tab;
    auto *m = new M256, *acc = new M256, *tmp = new M256;
    for (int b = 0; b < 256; b++) {
        V256 e{{0, 0, 0, 0}};
        e.w[b >> 6] = 1ull << (b & 63);
        m->col[b] = host_xo_step(e);
        acc->col[b] = e;  // identity
    }
    for (uint32_t e = 63u * SEG; e; e >>= 1) {  // acc = M^(63*SEG) by square and multiply
        if (e & 1u) {
            m_mul(*m, *acc, *tmp);
            *acc = *tmp;
        }
        m_mul(*m, *m, *tmp);
        *m = *tmp;
    }
    tab.resize((size_t)32 * 256 * 4);
    for (int k = 0; k < 32; k++)
        for (int v = 0; v < 256; v++) {
            V256 x{{0, 0, 0, 0}};
            x.w[k >> 3] = (uint64_t)v << (8 * (k & 7));
            const V256 r = m_apply(*acc, x);
            for (int q = 0; q < 4; q++) tab[((size_t)k * 256 + v) * 4 + q] = r.w[q];
        }
    delete m;
    delete acc;
    delete tmp;
    return tab;
}
const std::vector<uint64_t> &jump_table_host() {  // sessions run on several host threads: initialised exactly once
    static const std::vector<uint64_t> tab = [] {  // [seg_log - 3]: the tables of M^(63 * 8) and M^(63 * 16), back to back
        std::vector<uint64_t> t = build_jump_table(1u << JTK_SEG_LOG_GENERAL);
        const std::vector<uint64_t> u = build_jump_table(1u << JTK_SEG_LOG_LIGHT);
        t.insert(t.end(), u.begin(), u.end());
        return t;
    }();
    return tab;
}
std::mutex g_jump_mutex;
bool g_jump_uploaded[64];  // per device ordinal
}  // namespace

// The table is a constant of the generator: uploaded once per device, synchronously, before the first chain kernel.
int mcmc_upload_jump_table(hipStream_t s) {
    (void)s;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    std::lock_guard<std::mutex> lock(g_jump_mutex);
    if (g_jump_uploaded[dev]) return 0;
    const std::vector<uint64_t> &tab = jump_table_host();
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_jump_tab), tab.data(), tab.size() * 8, 0, hipMemcpyHostToDevice);
    if (e == hipSuccess) g_jump_uploaded[dev] = true;
    return (int)e;
}

// `split`: 2 + 2 * n_chunks words of device scratch, or null.  With it the launch is two kernels: the light one (168
// registers, the diploid chunks of <= 63 reads with <= 2 variant columns) and the general one for the rest; without it (or
// with JTK_MCMC_SPLIT=0) the general kernel runs everything.  `side` (with its two events), if given, is a second stream the
// general kernel runs on, beside the light one instead of before it.
int launch_mcmc(hipStream_t s, uint32_t n_chunks, const ChunkMeta *chunks, ChunkState *state,
                const jtk_lc_params_t *params, const double *feat, const uint32_t *vtype, const uint64_t *vt_off,
                uint32_t vt_stride_mode, uint32_t *label, double *post, uint32_t post_stride, double *lg,
                const uint64_t *lg_off, uint32_t lds_n, uint32_t lds_d, uint32_t lds_k, const uint64_t *rng_resume,
                const uint32_t *order, uint32_t *split, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join) {
    if (n_chunks == 0) return 0;
    lds_k = clamp_k(lds_k);
    const size_t lds = mcmc_lds_core(lds_n, lds_d, lds_k, JTK_SEG_LOG_GENERAL);  // the general kernel: a 12 KiB ring
    if (mcmc_upload_jump_table(s) != 0) return -1;  // the caller fails the call: nothing was launched
    const uint32_t flags = 0u;  // (reserved)
    static const bool no_split = getenv("JTK_MCMC_SPLIT") && atoi(getenv("JTK_MCMC_SPLIT")) == 0;
    if (!split || no_split || flags || rng_resume) {
        mcmc_kernel<<<n_chunks, 128, lds, s>>>(chunks, state, params, feat, vtype, vt_off, vt_stride_mode, label, post,
                                              post_stride, lg, lg_off, lds_n, lds_d, lds_k,
                                              JTK_SEG_LOG_GENERAL, flags, rng_resume, order, nullptr);
        return 0;
    }
    chain_split_kernel<<<1, 64, 0, s>>>(n_chunks, order, chunks, state, split);
    hipStream_t hs = s;
    static const bool no_side = getenv("JTK_MCMC_SIDE") && atoi(getenv("JTK_MCMC_SIDE")) == 0;  // experiments: one stream
    if (!no_side && side && ev_fork && ev_join && hipEventRecord(ev_fork, s) == hipSuccess &&
        hipStreamWaitEvent(side, ev_fork, 0) == hipSuccess)
        hs = side;
    mcmc_kernel<<<n_chunks, 128, lds, hs>>>(chunks, state, params, feat, vtype, vt_off, vt_stride_mode, label, post, post_stride,
                                           lg, lg_off, lds_n, lds_d, lds_k, JTK_SEG_LOG_GENERAL, 0u,
                                           nullptr, split + 2 + n_chunks, split + 1);
    const uint32_t ln = std::min<uint32_t>(lds_n, JTK_LIGHT_MAX_READS), ld = std::min<uint32_t>(lds_d, JTK_LIGHT_MAX_DIM);
    mcmc_kernel_light<<<n_chunks, 128, mcmc_lds_bytes(ln, ld, 2), s>>>(chunks, state, params, feat, vtype, vt_off, vt_stride_mode,
                                                                       label, post, post_stride, lg, lg_off, ln, ld, 2u,
                                                                       JTK_SEG_LOG_LIGHT, 0u, nullptr,
                                                                       split + 2, split);
    if (hs != s) {
        if (hipEventRecord(ev_join, hs) != hipSuccess || hipStreamWaitEvent(s, ev_join, 0) != hipSuccess) return -1;
    }
    return 0;
}

// The chunks listed in `order` (one workgroup each) through mcmc_kernel_huge; ws_off[i] = offset of order[i]'s slice of `ws`
// (mcmc_ws_bytes of ITS reads, columns and copy number).
int launch_mcmc_huge(hipStream_t s, uint32_t n_chunks, const ChunkMeta *chunks, ChunkState *state, const jtk_lc_params_t *params,
                     const double *feat, const uint32_t *vtype, const uint64_t *vt_off, uint32_t vt_stride_mode, uint32_t *label,
                     double *post, uint32_t post_stride, double *lg, const uint64_t *lg_off, uint32_t max_n, uint32_t max_d,
                     uint32_t max_k, const uint64_t *rng_resume, const uint32_t *order, unsigned char *ws, const uint64_t *ws_off) {
    if (n_chunks == 0) return 0;
    if (mcmc_upload_jump_table(s) != 0) return -1;
    mcmc_kernel_huge<<<n_chunks, 128, mcmc_lds_fixed() + JTK_HUGE_LDS, s>>>(chunks, state, params, feat, vtype, vt_off, vt_stride_mode, label, post,
                                                             post_stride, lg, lg_off, max_n, max_d, clamp_k(max_k), JTK_SEG_LOG_LIGHT, 0u,
                                                             rng_resume, order, nullptr, ws, ws_off);
    return 0;
}

// jtk_lc_session_trace: the chunk listed in order[0] once more through mcmc_kernel_trace (its results come out as they were);
// `trace` = mcmc_trace_doubles(n) doubles, `ws` = mcmc_ws_bytes of the chunk, ws_off[0] = 0.
size_t mcmc_trace_doubles(uint32_t n_reads) { return (size_t)JTK_TRACE_OLD + n_reads; }
int launch_mcmc_trace(hipStream_t s, const ChunkMeta *chunks, ChunkState *state, const jtk_lc_params_t *params, const double *feat,
                      const uint32_t *vtype, uint32_t *label, double *post, uint32_t post_stride, double *lg, const uint64_t *lg_off,
                      uint32_t n, uint32_t d, uint32_t k, const uint32_t *order, unsigned char *ws, const uint64_t *ws_off,
                      double *trace) {
    if (mcmc_upload_jump_table(s) != 0) return -1;
    mcmc_kernel_trace<<<1, 128, mcmc_lds_fixed() + JTK_HUGE_LDS, s>>>(chunks, state, params, feat, vtype, nullptr, 0u, label, post, post_stride,
                                                                     lg, lg_off, n, d, clamp_k(k), JTK_SEG_LOG_LIGHT, 0u, nullptr,
                                                                     order, nullptr, ws, ws_off, trace);
    return 0;
}
