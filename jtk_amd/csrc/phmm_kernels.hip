// phmm_kernels.hip -- band preparation and table finalisation of the banded pair-HMM for gfx950 (CDNA4); the sweep itself
// (phmm_kernel) lives in phmm_sweep.hip.  The round-2 sweep is kept below for differential runs only: a build with
// -DJTK_PHMM_WITH_R2 compiles it as phmm_kernel_r2 and JTK_PHMM_R2=1 selects it at run time; it is not in the product build.
//
// Replaces kiley `modification_table_antidiagonal` as called from
// haplotyper/src/local_clustering/pseudo_mcmc.rs:45-68 and the per-read inner step of
// `polish_until_converge_antidiagonal` (local_clustering/mod.rs:105-106).  kiley is not under
// /root/reference; the arithmetic implemented here is the own specification stated in DESIGN.md
// ("Pair-HMM specification") and checked bit-for-bit against oracle/phmm.c.
//
// Mapping to the hardware (one wavefront per read, 64 lanes):
//  * an anti-diagonal t = i + j of the banded DP is one wave-wide step; lane l owns the template row
//    i == l (mod 64) that lies in [c[t]-r, c[t]-r+63] ("lane ring").  With r <= 30 the band has <= 61
//    cells, the 3 spare lanes always hold zeros, so "value of row i-1" is a plain wave rotate by one lane
//    (v_mov_b32 dpp wave_ror:1 / wave_rol:1) and no per-step frame shift exists;
//  * template codes, the read (as emission indices) and both emission tables sit in LDS;
//  * the forward sweep streams, for every anti-diagonal s, the pair P_s = (toM of diagonal s-1, toD of diagonal s)
//    to a per-wave scratch stripe in HBM as one coalesced 1 KiB store (16 B per lane); the backward sweep reads
//    the pairs back through a register prefetch queue into an 8-slot LDS ring, from which the 16 row-crossing
//    products of a step take their operands as 7 ds_read_b128 at immediate offsets from one base register;
//  * each lane accumulates the 16 partial sums of ITS template row over time (no cross-lane reduction);
//    a row that leaves the band is flushed as 8 x 16 B stores; logs are taken later by finalize_kernel
//    with every lane busy;
//  * all scaling is by exact powers of two, one exponent per 64 anti-diagonals, so device and oracle agree
//    bit for bit as long as both use fma where the spec says fma (this file is built -ffp-contract=off).
#include <type_traits>

#include "device_common.h"

namespace {

__device__ __forceinline__ double rot_from_prev(double v) {  // lane l <- lane (l-1)&63
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x13C, 0xF, 0xF, false);  // wave_ror:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x13C, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rot_from_next(double v) {  // lane l <- lane (l+1)&63
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x134, 0xF, 0xF, false);  // wave_rol:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x134, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}
__device__ __forceinline__ double pow2i(int e) { return jtk_scalbn(1.0, e); }
// 2^e for the exponent differences of neighbouring scaling blocks (|e| is a few hundred at most); identical to
// scalbn(1.0, e) wherever the result is a normal number, which is the only case the fast branch accepts
__device__ __forceinline__ double fast_pow2(int e) {
    if (e == 0) return 1.0;
    if (e > -1000 && e < 1000) return jtk_bits_f64((uint64_t)(1023 + e) << 52);
    return jtk_scalbn(1.0, e);
}
__device__ __forceinline__ int delta_bit(const uint64_t *delta, int t) {  // c[t] - c[t-1], t >= 1
    const uint64_t w = delta[t >> 6];
    const uint32_t half = __builtin_amdgcn_readfirstlane((uint32_t)(w >> (t & 32)));
    return (int)((half >> (t & 31)) & 1u);
}

// ------------------------------------------------------------------------------------------------------
// band_prep: one thread per read walks its ops and writes delta bits (c[t] - c[t-1]); validates that the
// ops consume exactly the template and the read.  A Match step visits c = i+1 on both of its diagonals.
// ------------------------------------------------------------------------------------------------------
__global__ void band_prep_kernel(uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                                 ChunkState *state, DevBufs bufs, uint64_t *delta, int only_active) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const ReadMeta rm = reads[r];
    ChunkState *st = &state[rm.chunk];
    if (st->status != 0) return;
    if (only_active && !st->active) return;
    const uint32_t L = st->tmpl_len, n = rm.read_len, T = L + n;
    const uint8_t *ops = bufs.ops[st->buf] + rm.ops_off;
    const uint32_t n_ops = bufs.ops_len[st->buf][r];
    uint64_t *d = delta + rm.delta_off;
    const uint32_t words = (T >> 6) + 2;
    for (uint32_t w = 0; w < words; w++) d[w] = 0;
    uint32_t i = 0, j = 0, t = 0;
    bool bad = false;
    // the interval [f_lo, f_hi] of diagonals whose WHOLE band lies inside the DP matrix (0 <= i <= L, 0 <= j <= n for every
    // band cell): c - r >= 0, c + r <= L, c + r <= t, c - r >= t - n.  c and t - c are non-decreasing in t, so the set is one
    // interval; phmm_kernel runs it under an EXEC mask that is the band and takes per-lane predicates outside of it.
    const int32_t rad = (int32_t)chunks[rm.chunk].radius;
    uint32_t f_lo = 1, f_hi = 0;
    auto visit = [&](uint32_t tt, uint32_t cc) {  // diagonal tt has centre cc
        const int32_t c = (int32_t)cc, td = (int32_t)tt;
        if (c - rad >= 0 && c + rad <= (int32_t)L && c + rad <= td && c - rad >= td - (int32_t)n) {
            if (f_hi < f_lo) f_lo = tt;
            f_hi = tt;
        }
    };
    // the walk is one thread per read and latency bound: fetch the ops 8 at a time (ops_off is 8-byte aligned)
    uint64_t chunk8 = 0;
    for (uint32_t k = 0; k < n_ops; k++) {
        if ((k & 7u) == 0) chunk8 = *reinterpret_cast<const uint64_t *>(ops + k);
        uint8_t op = (uint8_t)(chunk8 >> (8 * (k & 7u)));
        if (op == JTK_OP_INS) {
            t += 1;
            j++;
            visit(t, i);
        } else if (op == JTK_OP_DEL) {
            t += 1;
            if (t <= T) d[t >> 6] |= 1ull << (t & 63);
            i++;
            visit(t, i);
        } else if (op <= JTK_OP_MISMATCH) {
            t += 1;
            if (t <= T) d[t >> 6] |= 1ull << (t & 63);
            i++;
            visit(t, i);
            t += 1;
            j++;
            visit(t, i);
        } else {
            bad = true;
        }
        if (t > T) {
            bad = true;
            break;
        }
    }
    if (bad || i != L || j != n) atomicMin(&st->status, (int)JTK_ERR_OPS_MISMATCH);
    // the word behind the read's delta words (session.hip sizes the slot from the template CAPACITY: one word to spare)
    d[((chunks[rm.chunk].tmpl_cap + rm.read_len) >> 6) + 2] = (bad || f_hi < f_lo) ? 1ull : ((uint64_t)f_hi << 32 | f_lo);
}

#ifdef JTK_PHMM_WITH_R2
// ------------------------------------------------------------------------------------------------------
// The per-read forward/backward sweep.
//
// Written for instruction count (with 11 waves per CU the kernel is bound by VALU/SALU issue, not by latency):
//  * a cell outside the band or outside the DP matrix holds exact zeros in every array, so neighbour terms need no
//    predicates of their own -- the only per-lane predicate of a step is "this cell exists" (`active`), plus the
//    one source row of the del-3 entry that the 3 spare lanes of the lane ring cannot disambiguate;
//    accumulator entries that finalize never reads (e.g. the sub/copy entries of row 0) may hold anything finite;
//  * the forward sweep stores, per anti-diagonal s, the pair P_s = (toM of diagonal s-1, toD of diagonal s) of the
//    lane's row: every one of the 16 row-crossing products of the backward sweep reads its (toM, toD) operands from
//    ONE such pair, so a step does 7 ds_read_b128 instead of 16 ds_read_b64;
//  * template / read codes are staged pre-multiplied into byte offsets of the emission tables and padded on both
//    sides, so the lookups need no clamping;
//  * the backward sweep is unrolled by 4 diagonals: prefetch registers and ring slots (up to one swapped base) are
//    compile-time; ring entries
//    are kept in the scale of the block that reads them, so the products carry no scale factors and the step has one
//    variant (the kernel is sensitive to its code size: the instruction cache is shared by two CUs).
// ------------------------------------------------------------------------------------------------------
#define PAD 64  // padding (bytes) in front of the staged code arrays; 64 more behind
#ifndef JTK_PHMM_PF
#define JTK_PHMM_PF 4  // pairs in flight from HBM per wave during the backward sweep (the unrolling assumes 4)
#endif
#ifndef JTK_PHMM_WAVES
#define JTK_PHMM_WAVES 3  // resident waves per SIMD the register budget is set for
#endif
#define RW 72   // entries per ring slot: 64 lanes + 4 wrapped copies in front + 2 behind (rounded up)

// Register budget: the 168 that three waves per SIMD allow (the attribute counts the unified file of gfx90a+ in halves,
// hence 84).  A chain wave holds 248 registers, so a SIMD that hosts one still takes a pair-HMM wave of another batch
// beside it (bench.py overlaps batches); while the chain kernel needed 360 this kernel was capped at 152 for that.
#ifndef JTK_PHMM_NUM_VGPR
#define JTK_PHMM_NUM_VGPR 84
#endif
__global__ __launch_bounds__(64, JTK_PHMM_WAVES) __attribute__((amdgpu_num_vgpr(JTK_PHMM_NUM_VGPR))) void phmm_kernel_r2(uint32_t n_reads, const ReadMeta *reads,
                                                  const ChunkMeta *chunks, const ChunkState *state,
                                                  DevBufs bufs, const uint8_t *ey_all, const uint64_t *delta_all,
                                                  const HmmDev *hmm2, double *scratch_all,
                                                  uint64_t scratch_stride, uint32_t *work_counter, double *raw_all,
                                                  int *rawG_all, double *lk_all, uint32_t lds_tmpl,
                                                  uint32_t lds_read, int only_active, uint32_t skip_le_radius) {
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS carve: ring [8][RW] double2 | eM[16] | eI[20] | delta words | block exponents | template codes | read codes
    // A ring slot holds the 64 lanes at entries 4..67 plus copies of lanes 60..63 in front and of lanes 0..1 behind,
    // so "the pair of row i+k" (k = -4..+2) is entry lane+4+k: one base register and immediate offsets.
    double2 *ring = reinterpret_cast<double2 *>(smem);
    unsigned char *s_eM = smem + 8 * RW * 16;  // doubles, addressed by byte offset
    unsigned char *s_eI = s_eM + 16 * 8;
    const uint32_t n_blk = ((lds_tmpl + lds_read) >> 6) + 4;
    uint64_t *s_delta = reinterpret_cast<uint64_t *>(s_eI + 20 * 8);  // band deltas of this read, one bit per diagonal
    int *s_EF = reinterpret_cast<int *>(s_delta + n_blk);
    uint8_t *s_xs = reinterpret_cast<uint8_t *>(s_EF + n_blk);  // s_xs[PAD + i - 1] = 32 * code(x[i-1]): row of eM in bytes
    uint8_t *s_ey = s_xs + ((lds_tmpl + 2 * PAD + 15) & ~15u);    // s_ey[PAD + j] = 8 * ey[j]: entry of eI in bytes
    const int lane = threadIdx.x;
    // the stripe starts with JTK_SCRATCH_GUARD rows of zeros: "the pair of a diagonal below 0" is then an ordinary load
    double2 *scratch = reinterpret_cast<double2 *>(scratch_all + (uint64_t)blockIdx.x * scratch_stride) + JTK_SCRATCH_GUARD * 64;
#pragma unroll
    for (int g = 1; g <= JTK_SCRATCH_GUARD; g++) scratch[-g * 64 + lane] = make_double2(0.0, 0.0);

    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(work_counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_reads) break;
        const ReadMeta rm = reads[item];
        const ChunkMeta cm = chunks[rm.chunk];
        const ChunkState st = state[rm.chunk];
        if (st.status != 0) continue;
        if (only_active && !st.active) continue;
        if (cm.take_num && item - cm.read_first >= cm.take_num) continue;  // this read does not vote (its ops are still re-threaded)
        if (cm.radius > JTK_MAX_RADIUS) continue;                          // phmm_wide_kernel's read
        if (cm.radius <= skip_le_radius) continue;                         // phmm_pair_kernel's read
        const int L = (int)st.tmpl_len, n = (int)rm.read_len, T = L + n, r = (int)cm.radius;
        const HmmDev *h = hmm2 + (rm.strand ? 0 : 1);
        const uint64_t *delta = delta_all + rm.delta_off;
        __syncthreads();
        // ---- stage the codes (as byte offsets, zero padded), the emission tables and the band deltas in LDS
        {
            const uint8_t *gx = bufs.tmpl[st.buf] + cm.tmpl_off;
            for (int p = lane; p < L + 2 * PAD; p += 64) {
                const int q = p - PAD;
                s_xs[p] = (q >= 0 && q < L) ? (uint8_t)(gx[q] << 5) : (uint8_t)0;
            }
            const uint8_t *gy = ey_all + rm.ey_off;
            for (int p = lane; p < n + 1 + 2 * PAD; p += 64) {
                const int q = p - PAD;
                s_ey[p] = (q >= 1 && q <= n) ? (uint8_t)(gy[q] << 3) : (uint8_t)0;
            }
            if (lane < 16) reinterpret_cast<double *>(s_eM)[lane] = h->eM[lane];
            if (lane < 20) reinterpret_cast<double *>(s_eI)[lane] = h->eI[lane];
            for (int wdx = lane; wdx < (T >> 6) + 2; wdx += 64) s_delta[wdx] = delta[wdx];
        }
        __syncthreads();
        const double aMM = h->a[0], aMI = h->a[1], aMD = h->a[2], aIM = h->a[3], aII = h->a[4], aID = h->a[5],
                     aDM = h->a[6], aDI = h->a[7], aDD = h->a[8];
        const uint8_t *xs0 = s_xs + PAD - 1;  // xs0[i] = row offset of x[i-1]
        const uint8_t *ey0 = s_ey + PAD;      // ey0[j] = entry offset of read base j

        // =========================== forward ===========================
        int c = 0, EF = 0;
        double toM_1 = 0, toM_2 = 0, toI_1 = 0, toD_1 = 0;  // combos of diagonals t-1 / t-2, lane frame
        double endM = 0, endI = 0, endD = 0;
        {  // t == 0: the only cell is (0, 0), on lane 0
            const double fm = lane == 0 ? 1.0 : 0.0;
            toM_1 = fm * aMM;
            toI_1 = fm * aMI;
            toD_1 = fm * aMD;
            scratch[lane] = make_double2(0.0, toD_1);
            if (lane == 0) s_EF[0] = 0;
            if (T == 0) endM = fm;
        }
        // The emission look-ups of a cell (two dependent LDS reads) do not depend on the recurrence, so they are
        // issued one anti-diagonal ahead: with two waves per SIMD nothing else would cover their latency.
        auto sdelta = [&](int w) -> uint64_t {  // a word of band deltas as a scalar
            const uint64_t v = s_delta[w];
            return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
        };
        uint64_t dw = sdelta(0);
        int c_n = 0;
        bool act_n = false;
        double eM_n = 0.0, eI_n = 0.0;
        auto prefetch_fwd = [&](int tn) {  // the cell of diagonal tn on this lane
            if ((tn & 63) == 0) dw = sdelta(tn >> 6);
            c_n += (int)((dw >> (tn & 63)) & 1ull);
            const int lo = c_n - r, off = (lane - lo) & 63, i = lo + off, j = tn - i;
            act_n = off <= 2 * r && (unsigned)i <= (unsigned)L && (unsigned)j <= (unsigned)n;
            const int ey8 = ey0[j], xs = xs0[i];
            eM_n = *reinterpret_cast<const double *>(s_eM + xs + (ey8 & 24));
            eI_n = *reinterpret_cast<const double *>(s_eI + ey8);
        };
        if (T >= 1) prefetch_fwd(1);
        for (int t = 1; t <= T; t++) {
            c = c_n;
            const bool active = act_n;
            const double eMv = eM_n, eIv = eI_n;
            if (t < T) prefetch_fwd(t + 1);
            const double pM = rot_from_prev(toM_2), pD = rot_from_prev(toD_1);
            double fm = eMv * pM, fi = eIv * toI_1, fd = pD;
            if (!active) fm = fi = fd = 0.0;
            const double toM_prev = toM_1;  // toM of diagonal t-1 in its own block's scale: what the pair stores
            if ((t & (JTK_SCALE_BLOCK - 1)) == 0) {
                double mx = fm > fi ? fm : fi;
                mx = fd > mx ? fd : mx;
                mx = wave_max(mx);
                if (mx > 0.0) {
                    const int e = __builtin_amdgcn_readfirstlane(jtk_ilogb_pos(mx));  // the same in every lane: keep it scalar
                    const double sc = pow2i(-e);
                    fm *= sc;
                    fi *= sc;
                    fd *= sc;
                    toM_1 *= sc;
                    EF += e;
                }
                if (lane == 0) s_EF[t >> 6] = EF;
            }
            const double toM = fma(fd, aDM, fma(fi, aIM, fm * aMM));
            const double toI = fma(fd, aDI, fma(fi, aII, fm * aMI));
            const double toD = fma(fd, aDD, fma(fi, aID, fm * aMD));
#ifndef JTK_PHMM_EXPERIMENT_NOSTORE
            scratch[(uint64_t)t * 64 + lane] = make_double2(toM_prev, toD);
#endif
            toM_2 = toM_1;
            toM_1 = toM;
            toI_1 = toI;
            toD_1 = toD;
            if (t == T) {
                endM = fm;
                endI = fi;
                endD = fd;
            }
        }
        scratch[(uint64_t)(T + 1) * 64 + lane] = make_double2(toM_1, 0.0);  // P_{T+1} = (toM of diagonal T, nothing)
        // cell (L, n) sits on the lane that owns row L
        const int lane_end = L & 63;
        double tot = (endM + endI) + endD;
        tot = __shfl(tot, lane_end, 64);
        const double lk = tot > 0.0 ? jtk_log(tot) + (double)EF * JTK_LN2 : JTK_LOG_ZERO;
        if (lane == 0) lk_all[item] = lk;
        __syncthreads();  // s_EF visible; forward stores are read back by this same wave below
#ifdef JTK_PHMM_EXPERIMENT_FWDONLY
        continue;
#endif

        // =========================== backward + table accumulation ===========================
        double *raw = raw_all + rm.raw_off;
        int *rawG = rawG_all + rm.row_off;
        double acc[JTK_ACC_N];
#pragma unroll
        for (int k = 0; k < JTK_ACC_N; k++) acc[k] = 0.0;
        double hM_1 = 0, hM_2 = 0, hI_1 = 0, bD_1 = 0;  // hatM(t+1), hatM(t+2), hatI(t+1), b_D(t+1)
        int EB = 0, Gprev = 0;
        // c == c[T]; centres of the del-3 source diagonals t-5 / t-4 are tracked separately
        int c5 = c, c4 = c;
        {
            int cc = c, tt = T;
            for (int k = 0; k < 4 && tt >= 1; k++, tt--) cc -= delta_bit(s_delta, tt);
            c4 = cc;
            if (tt >= 1) cc -= delta_bit(s_delta, tt);
            c5 = cc;
        }
        auto load_pair = [&](int ss) -> double2 {
#ifdef JTK_PHMM_EXPERIMENT_NOLOAD
            return make_double2(1e-3 * ss, 0.5);
#else
            return scratch[(int64_t)ss * 64 + lane];  // ss >= -JTK_SCRATCH_GUARD
#endif
        };
        // ring: P_{T+2} (nothing) .. P_{T-4}; queue: pq[s & (PF-1)] = P_s for the next PF below
        // the wrapped copies without exec-mask branches: lanes 0..1 / 60..63 write their copy, the others their own entry again
        const int wrap_off = lane < 2 ? 64 : (lane >= 60 ? -64 : 0);
        auto ring_put_at = [&](double2 *e, double2 v) __attribute__((always_inline)) {  // e: the lane's entry of a slot
            e[0] = v;
            e[wrap_off] = v;
        };
        auto ring_put = [&](int slot, double2 v) __attribute__((always_inline)) { ring_put_at(ring + slot * RW + lane + 4, v); };
        // Ring entries are kept in the scale of the block of the step that reads them: a pair enters multiplied by the
        // exact power of two between its diagonals' blocks and the current one, and when the sweep crosses into the
        // block below, everything in the ring is re-expressed once.  The products then carry no scale factors (and the
        // backward step exists in one variant: this kernel is sensitive to its code size).
        auto rel = [&](int d, int blk) -> double {  // 2^(EF[block of diagonal d] - EF[blk])
            if (d < 0 || d > T) return 1.0;
            return fast_pow2(s_EF[d >> 6] - s_EF[blk]);
        };
        ring_put((T + 2) & 7, make_double2(0.0, 0.0));
        for (int ss = T + 1; ss >= T - 4; ss--) {
            double2 v = load_pair(ss);
            v.x *= rel(ss - 1, T >> 6);
            v.y *= rel(ss, T >> 6);
            ring_put(ss & 7, v);
        }
        // The pairs of a group of four diagonals are loaded as one batch a whole group ahead (pqY) and handed over at the
        // group's end (pq = pqY): the flush's conditional stores make the compiler wait for EVERY outstanding load
        // (s_waitcnt vmcnt(0)) wherever a loaded register is used; with one load per step each load had a single step to
        // arrive -- a batch issued after the group's first use has four.  pq[idx] holds the pair P_s of the current group
        // with s == idx (mod 4).
        auto s_of = [&](int tb, int idx) -> int { return tb - 5 - ((2 - idx) & 3); };  // the s in {tb-8 .. tb-5} with s == idx (mod 4)
        double2 pq[JTK_PHMM_PF], pqY[JTK_PHMM_PF];
#pragma unroll
        for (int q = 0; q < JTK_PHMM_PF; q++) pq[q] = load_pair(s_of(T | 3, q));
        int delta_next = 0;  // c[t+1] - c[t]
        int EFcur = __builtin_amdgcn_readfirstlane(s_EF[T >> 6]);  // forward exponent of the block the sweep is in (scalar)
        const double2 *ring_me = ring + lane + 4;  // source row i+k: entry ring_me[k] of its slot

        // One backward step.  The sweep is unrolled by 4 (t & 3 == 3 - U inside a group), which makes the queue
        // register and the low two bits of every ring slot compile-time; bit 2 of a slot follows bit 2 of its
        // diagonal, i.e. (t >> 2) & 1 plus a compile-time carry: half[0] / half[1] are the lane's entries in the
        // slot halves {0..3} / {4..7} for even / odd carry and swap from group to group.
        double2 *half[2] = {ring + lane + 4, ring + lane + 4 + 4 * RW};
        auto step = [&](int t, auto u_tag, auto pq_tag) __attribute__((always_inline)) {
            constexpr int pq_idx = decltype(pq_tag)::value;
            constexpr int U = decltype(u_tag)::value;
            auto entry = [&](int dx) -> double2 * {  // the lane's entry of the slot of pair P_{t+dx}; dx is a literal
                const int lo2 = (3 - U) + dx;         // (t & 3) + dx
                return half[(lo2 >> 2) & 1] + (lo2 & 3) * RW;
            };
            if (U == 0 && (t & 63) == 63 && t < T) {  // the sweep enters the block below
                const int EFabove = EFcur;
                EFcur = __builtin_amdgcn_readfirstlane(s_EF[t >> 6]);
                const double f = fast_pow2(EFabove - EFcur);
#pragma unroll
                for (int sl = 0; sl < 8; sl++) {
                    double2 v = ring_me[sl * RW];
                    v.x *= f;
                    v.y *= f;
                    ring_put(sl, v);
                }
            }
            if (t < T) c -= delta_next;  // centres: c == c[t+1] on entry
            const int lo = c - r, off = (lane - lo) & 63, i = lo + off, j = t - i;
            const bool active = off <= 2 * r && (unsigned)i <= (unsigned)L && (unsigned)j <= (unsigned)n;
            // (0) a row that left the band at this step is final (exponent of the previous step)
            if (t < T && delta_next == 1) {
                if (off == 2 * r + 1 && (unsigned)i <= (unsigned)L) {
                    double2 *dst = reinterpret_cast<double2 *>(raw + (uint64_t)i * JTK_ACC_N);
#pragma unroll
                    for (int k = 0; k < JTK_ACC_N / 2; k++) dst[k] = make_double2(acc[2 * k], acc[2 * k + 1]);
                    rawG[i] = Gprev;
#pragma unroll
                    for (int k = 0; k < JTK_ACC_N; k++) acc[k] = 0.0;
                }
            }
            // the pair five diagonals below enters the ring now (its slot is not read by this step)
            {
                double2 v = pq[pq_idx];
                if ((t & 63) <= 5) {  // its diagonals t-6 / t-5 lie in the block below
                    v.x *= rel(t - 6, t >> 6);
                    v.y *= rel(t - 5, t >> 6);
                }
                ring_put_at(entry(-5), v);
            }
            // (1) backward values of this diagonal
            double vm, vi, vd;
            if (t == T) {
                vm = vi = vd = 1.0;
            } else {
                const double xm = rot_from_next(hM_2), xd = rot_from_next(bD_1), xi = hI_1;
                vm = fma(aMD, xd, fma(aMI, xi, aMM * xm));
                vi = fma(aID, xd, fma(aII, xi, aIM * xm));
                vd = fma(aDD, xd, fma(aDI, xi, aDM * xm));
            }
            if (!active) vm = vi = vd = 0.0;
            if (t < T && (t & (JTK_SCALE_BLOCK - 1)) == JTK_SCALE_BLOCK - 1) {
                double mx = vm > vi ? vm : vi;
                mx = vd > mx ? vd : mx;
                mx = wave_max(mx);
                if (mx > 0.0) {
                    const int e = __builtin_amdgcn_readfirstlane(jtk_ilogb_pos(mx));
                    const double sc = pow2i(-e);
                    vm *= sc;
                    vi *= sc;
                    vd *= sc;
                    hM_1 *= sc;
                    EB += e;
                }
            }
            const int ey8 = ey0[j], xs = xs0[i], y8 = ey8 & 24;
            const double hM = *reinterpret_cast<const double *>(s_eM + xs + y8) * vm;
            const double hI = *reinterpret_cast<const double *>(s_eI + ey8) * vi;
            // (2) common exponent of this step
            const int G = EFcur + EB;
            if (t < T && G != Gprev) {
                const double sc = pow2i(Gprev - G);
#pragma unroll
                for (int k = 0; k < JTK_ACC_N; k++) acc[k] *= sc;
            }
            Gprev = G;
            // (3) the 16 row-crossing products of this cell.  Pair P_x holds toM of diagonal x-1 and toD of diagonal x,
            //     already in this step's scale.
            auto pair = [&](int x, int k) -> double2 {  // x: diagonal offset from t (a literal after unrolling)
#ifdef JTK_PHMM_EXPERIMENT_NORING
                return make_double2(1e-3 * k + hM_1, 0.25 + bD_1);
#else
                return entry(x)[k];
#endif
            };
            // vm enters the accumulator of this cell's read base only: one masked copy per base, shared by the sub and
            // the ins entries (fma(x, 0, acc) == acc exactly, so masking vm instead of the pair changes no bit)
            double vmq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) vmq[q] = y8 == 8 * q ? vm : 0.0;
            {  // sub (entry i-1): toM(i-1, j-1), toD(i-1, j)
                const double2 a = pair(-1, -1);
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] = fma(a.x, vmq[q], acc[q]);
                acc[4] = fma(a.y, vd, acc[4]);
            }
            {  // ins (entry i): toM(i, j-1), toD(i, j);   copy 1 (entry i-1): the same pair against hatM
                const double2 a = pair(0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) acc[5 + q] = fma(a.x, vmq[q], acc[5 + q]);
                acc[9] = fma(a.y, vd, acc[9]);
                acc[10] = fma(a.y, vd, fma(a.x, hM, acc[10]));
            }
#pragma unroll
            for (int cc = 2; cc <= 3; cc++) {  // copy c (entry i-1): toM(i-1+c, j-1), toD(i-1+c, j)
                const double2 a = pair(cc - 1, cc - 1);
                acc[10 + cc - 1] = fma(a.y, vd, fma(a.x, hM, acc[10 + cc - 1]));
            }
#pragma unroll
            for (int dd = 1; dd <= 3; dd++) {  // del d (entry i-d-1): toM(i-d-1, j-1), toD(i-d-1, j)
                double2 a = pair(-dd - 1, -dd - 1);
                if (dd == 3) {  // the only source row the 3 spare lanes cannot disambiguate
                    if (!(i - 4 >= c5 - r)) a.x = 0.0;
                    if (!(i - 4 >= c4 - r)) a.y = 0.0;
                }
                acc[13 + dd - 1] = fma(a.y, vd, fma(a.x, hM, acc[13 + dd - 1]));
            }
            hM_2 = hM_1;
            hM_1 = hM;
            hI_1 = hI;
            bD_1 = vd;
            // centres for the next step (t-1): c[t] -> c[t-1], c5 = c[t-6], c4 = c[t-5]
            delta_next = t >= 1 ? delta_bit(s_delta, t) : 0;
            c4 = c5;
            if (t - 5 >= 1) c5 -= delta_bit(s_delta, t - 5);
        };
        // groups of 4 diagonals, tb == 3 (mod 4)
        for (int tb = T | 3; tb >= 3; tb -= 4) {
            {
                double2 *lo_half = ring + lane + 4, *hi_half = lo_half + 4 * RW;
                const bool hi = (tb >> 2) & 1;
                half[0] = hi ? hi_half : lo_half;
                half[1] = hi ? lo_half : hi_half;
            }
#define GROUP_STEP(u)                                                                                      \
    {                                                                                                      \
        const int t = tb - (u);                                                                            \
        if (t <= T) step(t, std::integral_constant<int, (u)>{}, std::integral_constant<int, (2 - (u)) & 3>{}); \
    }
            GROUP_STEP(0)
            // the next group's pairs, issued right after this group's first use of a loaded pair: that use is where the
            // compiler waits for everything outstanding, so these loads are not waited for until the next group
#pragma unroll
            for (int q = 0; q < JTK_PHMM_PF; q++) pqY[q] = load_pair(s_of(tb, q) - 4);
            GROUP_STEP(1)
            GROUP_STEP(2)
            GROUP_STEP(3)
#undef GROUP_STEP
#pragma unroll
            for (int q = 0; q < JTK_PHMM_PF; q++) pq[q] = pqY[q];
        }
        // rows still in the band after t == 0
        {
            const int lo = c - r, off = (lane - lo) & 63, i = lo + off;
            if (off <= 2 * r && (unsigned)i <= (unsigned)L) {
                double2 *dst = reinterpret_cast<double2 *>(raw + (uint64_t)i * JTK_ACC_N);
#pragma unroll
                for (int k = 0; k < JTK_ACC_N / 2; k++) dst[k] = make_double2(acc[2 * k], acc[2 * k + 1]);
                rawG[i] = Gprev;
            }
        }
    }
}

#endif  // JTK_PHMM_WITH_R2

// ------------------------------------------------------------------------------------------------------
// finalize: the 14 table entries of every position p of a read from its raw row sums, MINUS the read's lk
// (pseudo_mcmc.rs:64).  Row iota owns sub[iota-1], ins[iota], copy_c[iota-1], del_d[iota-d-1].
//
// IN PLACE: the table of a read (14 doubles per position) takes the place of its row sums (16 per position, the same
// region of HBM): position p reads rows p .. p+4 and its 14 entries land in [14p, 14p+14) -- inside rows <= p, which no later
// position reads.  One workgroup walks a read's tiles of 128 positions in order (a tile's rows are staged in LDS before
// anything of the tile is written, and its writes end below the first row of the next tile), so the per-read table costs no
// memory of its own: 9.7 GB less per 625-chunk slice of the headline workload.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double fin_log(double v, int G, double lk) {
    return (v > 0.0 ? jtk_log(v) + (double)G * JTK_LN2 : JTK_LOG_ZERO) - lk;
}

#define FIN_TILE 128                   // positions per tile
#define FIN_ROWS (FIN_TILE + 4)        // raw rows a tile reads: p .. p+4 for its last position
#define FIN_PITCH (JTK_ACC_N + 1)      // doubles per staged row: 17 keeps the 128-byte rows off each other's LDS banks
__global__ __launch_bounds__(FIN_TILE) void finalize_kernel(uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                                                           const ChunkState *state, const HmmDev *hmm2, double *raw_all,
                                                           const int *rawG_all, const double *lk_all, int only_active) {
    // A tile's raw rows come in through LDS with coalesced 16-byte loads and its 14 x 128 table entries leave the same way:
    // a thread's own rows (128 B apart) and its 112 bytes of output would otherwise be 16-byte pieces of different lines.
    __shared__ __align__(16) double s_raw[FIN_ROWS * FIN_PITCH];
    __shared__ int s_G[FIN_ROWS];
    const uint32_t item = blockIdx.x;
    const ReadMeta rm = reads[item];
    const ChunkState st = state[rm.chunk];
    if (st.status != 0) return;
    if (only_active && !st.active) return;
    {
        const ChunkMeta &cm = chunks[rm.chunk];
        if (cm.take_num && item - cm.read_first >= cm.take_num) return;
    }
    const int L = (int)st.tmpl_len;
    const int tid = threadIdx.x;
    const HmmDev *h = hmm2 + (rm.strand ? 0 : 1);
    double *raw = raw_all + rm.raw_off;  // == the read's table (ReadMeta.table_off == raw_off)
    const int *rawG = rawG_all + rm.row_off;
    const double lk = lk_all[item];
    const bool dead = !(lk > JTK_LOG_ZERO);
    double eM[16];
#pragma unroll
    for (int k = 0; k < 16; k++) eM[k] = h->eM[k];
    for (int p0 = 0; p0 <= L; p0 += FIN_TILE) {
        const int p = p0 + tid;
        const int n_rows = min(FIN_ROWS, L + 1 - p0);  // rows p0 .. min(p0 + FIN_ROWS - 1, L)
        if (!dead) {
            const double2 *src = reinterpret_cast<const double2 *>(raw + (uint64_t)p0 * JTK_ACC_N);
            for (int e = tid; e < n_rows * (JTK_ACC_N / 2); e += FIN_TILE) {
                const double2 v = src[e];
                const int row = e / (JTK_ACC_N / 2), k = e % (JTK_ACC_N / 2);
                s_raw[row * FIN_PITCH + 2 * k] = v.x;
                s_raw[row * FIN_PITCH + 2 * k + 1] = v.y;
            }
            for (int e = tid; e < n_rows; e += FIN_TILE) s_G[e] = rawG[p0 + e];
        }
        __syncthreads();
        double res[JTK_NUM_ROW];
#pragma unroll
        for (int k = 0; k < JTK_NUM_ROW; k++) res[k] = JTK_LOG_ZERO - (dead ? 0.0 : lk);
        if (!dead && p <= L) {
            if (p + 1 <= L) {  // row p+1: sub[p], copy_c[p]
                const double *a = s_raw + (tid + 1) * FIN_PITCH;
                const int G = s_G[tid + 1];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    double v = eM[4 * b + 0] * a[0];
                    v = fma(eM[4 * b + 1], a[1], v);
                    v = fma(eM[4 * b + 2], a[2], v);
                    v = fma(eM[4 * b + 3], a[3], v);
                    v = v + a[4];
                    res[b] = fin_log(v, G, lk);
                }
#pragma unroll
                for (int cc = 0; cc < 3; cc++) res[8 + cc] = fin_log(a[10 + cc], G, lk);
            }
            {  // row p: ins[p]
                const double *a = s_raw + tid * FIN_PITCH;
                const int G = s_G[tid];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    double v = eM[4 * b + 0] * a[5];
                    v = fma(eM[4 * b + 1], a[6], v);
                    v = fma(eM[4 * b + 2], a[7], v);
                    v = fma(eM[4 * b + 3], a[8], v);
                    v = v + a[9];
                    res[4 + b] = fin_log(v, G, lk);
                }
            }
#pragma unroll
            for (int dd = 1; dd <= 3; dd++) {  // row p+d+1: del_d[p]
                if (p + dd + 1 <= L) {
                    const double *a = s_raw + (tid + dd + 1) * FIN_PITCH;
                    res[11 + dd - 1] = fin_log(a[13 + dd - 1], s_G[tid + dd + 1], lk);
                }
            }
        }
        __syncthreads();  // the staged rows are not needed any more: the tile's table entries take their place
        double *s_out = s_raw;  // FIN_TILE x 14 doubles <= FIN_ROWS x 17
        if (p <= L) {
#pragma unroll
            for (int k = 0; k < JTK_NUM_ROW; k++) s_out[tid * JTK_NUM_ROW + k] = res[k];
        }
        __syncthreads();
        const int n_pos = min(FIN_TILE, L + 1 - p0);
        double2 *dst = reinterpret_cast<double2 *>(raw + (uint64_t)p0 * JTK_NUM_ROW);
        const double2 *so = reinterpret_cast<const double2 *>(s_out);
        for (int e = tid; e < n_pos * (JTK_NUM_ROW / 2); e += FIN_TILE) dst[e] = so[e];
        __syncthreads();  // before the next tile is staged over s_out
    }
}

}  // namespace

void launch_band_prep(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                      ChunkState *state, DevBufs bufs, uint64_t *delta, int only_active) {
    if (n_reads == 0) return;
    band_prep_kernel<<<(n_reads + 63) / 64, 64, 0, s>>>(n_reads, reads, chunks, state, bufs, delta, only_active);
}

#ifdef JTK_PHMM_WITH_R2
static size_t phmm_r2_lds_bytes(uint32_t max_tmpl, uint32_t max_read) {
    const uint32_t n_blk = ((max_tmpl + max_read) >> 6) + 4;
    size_t b = 8 * RW * 16 + 36 * 8 + (size_t)n_blk * 12;
    b += ((max_tmpl + 2 * PAD + 15) & ~15u) + max_read + 1 + 2 * PAD;
    return (b + 15) & ~(size_t)15;
}
void launch_phmm_r2(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                    const ChunkState *state, DevBufs bufs, const uint8_t *ey, const uint64_t *delta,
                    const HmmDev *hmm2, double *scratch, uint64_t scratch_stride, uint32_t n_waves,
                    uint32_t *work_counter, double *raw, int *rawG, double *lk, uint32_t max_tmpl,
                    uint32_t max_read, int only_active, uint32_t skip_le_radius) {
    if (n_reads == 0) return;
    (void)hipMemsetAsync(work_counter, 0, sizeof(uint32_t), s);
    phmm_kernel_r2<<<n_waves, 64, phmm_r2_lds_bytes(max_tmpl, max_read), s>>>(n_reads, reads, chunks, state, bufs, ey, delta, hmm2,
                                                                             scratch, scratch_stride, work_counter, raw, rawG, lk,
                                                                             max_tmpl, max_read, only_active, skip_le_radius);
}
#endif

void launch_finalize(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                     const ChunkState *state, const HmmDev *hmm2, double *raw, const int *rawG, const double *lk,
                     uint32_t max_tmpl, int only_active) {
    if (n_reads == 0) return;
    (void)max_tmpl;
    finalize_kernel<<<n_reads, FIN_TILE, 0, s>>>(n_reads, reads, chunks, state, hmm2, raw, rawG, lk, only_active);
}
