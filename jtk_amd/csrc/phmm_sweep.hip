// phmm_sweep.hip -- phmm_kernel: banded pair-HMM forward/backward + modification-table row sums for band radius 15..30
// (ONT at 2 kbp: 30), one wavefront per read, gfx950.
//
// Replaces kiley `modification_table_antidiagonal` (haplotyper/src/local_clustering/pseudo_mcmc.rs:45-68) and the per-read
// inner step of `polish_until_converge_antidiagonal` (local_clustering/mod.rs:105-106).  kiley is not under /root/reference;
// the arithmetic is the own specification of DESIGN.md section 4 / oracle/phmm.c, reproduced bit for bit.
//
// Mapping (round 3 rewrite; the round-2 kernel issued 290 wave-instructions per anti-diagonal, profiles/r03_pmc_issue_phmm_base.txt):
//  * lane ring as before: an anti-diagonal t = i + j is one wave-wide step, lane l owns the template row i == l (mod 64)
//    inside [c[t]-r, c[t]-r+63]; <= 61 band cells + 3 spare lanes, neighbours are wave_ror/rol:1 DPP moves;
//  * the MIDDLE of both sweeps (all band cells inside the DP matrix: band_prep_kernel computes that interval per read) runs in
//    groups of 8 unrolled diagonals under an EXEC mask that is the band itself (a rotating 64-bit scalar mask,
//    __builtin_amdgcn_inverse_ballot_w64): no per-lane "does this cell exist" arithmetic, no selects; spare lanes keep
//    exact zeros because they simply do not execute.  A lane's template row changes only when the band moves past it, so
//    row index, emission row and the lane's read-byte window are registers updated by ONE lane when a row leaves the band;
//  * the read's emission bytes of a group's 8 diagonals are two dwords per lane (3 aligned LDS dwords + v_alignbyte);
//  * band deltas, scaling exponents and the del-3 thresholds are scalar (one LDS byte pair per group);
//  * row sums of a row that left the band stay in the (now idle) lane and are flushed once per HALF group (4 diagonals) for
//    all lanes that left: a lane that leaves the band re-enters it with its next row at the third later band move, so a half
//    group may hold at most 3 band moves (its read-byte window and its staged rows are per half group for the same reason);
//  * the first / last ~2r diagonals of a sweep, and any group with a half of 4 band moves, take a generic step (per-lane
//    predicates, as the round-2 kernel did everywhere);
//  * the pair P_s = (toM of diagonal s-1, toD of diagonal s) the backward sweep needs of every diagonal goes through an HBM
//    stripe (1 KiB per diagonal) only at the ends of the sweeps.  Round 4: wherever the backward sweep runs a fast group
//    (diagonals 8g .. 8g+7) it needs the pairs of the diagonals 8g-5 .. 8g+2, and instead of 8 KiB written and read back it gets
//    ONE checkpoint of the forward state after diagonal 8g-6 (toM_1, toM_2, toI_1, toD_1: 2 KiB) and REPLAYS the eight forward
//    steps in front of its own eight (the same instructions on the same values: the same bits); the replayed pairs wait in the
//    registers the prefetch queue used to occupy and enter the 8-slot LDS ring one per step as before.  Which groups replay is
//    a bit per group, worked out once per read from the band deltas (bf bits: the backward group is fast AND both forward
//    groups that produce its pairs were).
#include "device_common.h"


namespace {

#define PAD 64                 // zero padding (bytes) either side of the staged code arrays
#define RS 1024                // bytes per ring slot: 64 lanes x (toM, toD)
#define S_EM (8 * RS)          // eM[16] as doubles (32-byte aligned: a row's address can be OR-ed with the column)
#define S_EI (S_EM + 128)      // eI[20]
#define S_STAGE (S_EI + 160)   // 3 finished rows x 16 row sums on their way out (a half group retires at most 3 rows)
#define S_SMETA (S_STAGE + 384)  // their (row, exponent)
#define S_VAR (S_SMETA + 32)   // delta words | block exponents | replay bits | template codes | read codes
#define JTK_BFW_BYTES(n_blk) ((n_blk) + 8u)  // one bit per group of 8 diagonals, n_blk blocks of 64 diagonals; + slack for the look-ahead
#ifndef JTK_PHMM_REPLAY
#define JTK_PHMM_REPLAY 1      // 0: every pair through the stripe (rounds 2-3)
#endif
#ifndef JTK_PHMM_FWD_PREFETCH
#define JTK_PHMM_FWD_PREFETCH 1     // the forward sweep's fast steps do the same (and load a half group's read bytes a step early)
#endif
#ifndef JTK_PHMM_BASE_CMPX
#define JTK_PHMM_BASE_CMPX 1        // the row sums split by read base through v_cmpx (0: v_cmp + s_and_saveexec, rounds 3-5)
#endif
#ifndef JTK_PHMM_REPLAY_PREFETCH
#define JTK_PHMM_REPLAY_PREFETCH 1  // the replayed steps fetch their emission entries one step ahead (0: when they need them)
#endif

__device__ __forceinline__ double rot_from_prev(double v) {  // lane l <- lane (l-1)&63
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x13C, 0xF, 0xF, false);  // wave_ror:1 (every lane has a source: no `old` value needed)
    hi = __builtin_amdgcn_mov_dpp(hi, 0x13C, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rot_from_next(double v) {  // lane l <- lane (l+1)&63
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x134, 0xF, 0xF, false);  // wave_rol:1
    hi = __builtin_amdgcn_mov_dpp(hi, 0x134, 0xF, 0xF, false);
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
__device__ __forceinline__ double fast_pow2(int e) {  // == scalbn(1.0, e) wherever the result is normal
    if (e == 0) return 1.0;
    if (e > -1000 && e < 1000) return jtk_bits_f64((uint64_t)(1023 + e) << 52);
    return jtk_scalbn(1.0, e);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    return (uint64_t)(uint32_t)uni((int)(uint32_t)(v >> 32)) << 32 | (uint32_t)uni((int)(uint32_t)v);
}
__device__ __forceinline__ double uni_f64(double v) { return jtk_bits_f64(uni64(jtk_f64_bits(v))); }
// v with lane l (uniform, a scalar register) replaced by the scalar x: v_writelane_b32 ignores EXEC, so one lane's state changes
// without a saveexec / branch / restore around it
__device__ __forceinline__ __attribute__((unused)) int wlane(int x, int l, int v) {
    x = __builtin_amdgcn_readfirstlane(x);  // (a uniform value the compiler keeps in a vector register is not an "s" operand)
    l = __builtin_amdgcn_readfirstlane(l);
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(x), "s"(l) : "m0");  // (two SGPRs: the select goes through M0)
    return v;
}
__device__ __forceinline__ __attribute__((unused)) double wl_zero(double v, int l) {  // v with lane l set to +0.0
    int lo = __double2loint(v), hi = __double2hiint(v);
    asm("v_writelane_b32 %0, 0, %2\n\tv_writelane_b32 %1, 0, %2" : "+v"(lo), "+v"(hi) : "s"(l));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint64_t rotl64(uint64_t m, int s) { return (m << (s & 63)) | (m >> ((64 - s) & 63)); }
__device__ __forceinline__ bool lanes(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// A block that must run under an EXEC mask (and not be turned into selects over everything it assigns): an empty volatile
// asm cannot be speculated, so the block stays a branch.
#define KEEP_MASKED asm volatile("")
#ifdef JTK_PHMM_MARKS  // region markers in the assembly, for reading it (the compiler lays blocks out of order: no census by text order)
#define MARK(x) asm volatile("; MARK " x)
#else
#define MARK(x)
#endif

// acc_sub += x_sub * vm and acc_ins += x_ins * vm on the lanes whose read base is `Q` (y8 == 8 Q): the row sums split by
// read base.  fma(x, 0, acc) == acc exactly, so leaving the other lanes out changes no bit (oracle/phmm.c masks vm).
#define BASE_FMA(Q, acc_sub, acc_ins, x_sub, x_ins, vmv, y8v)                                                         \
    {                                                                                                               \
        uint64_t sv_;                                                                                               \
        asm("v_cmp_eq_u32_e32 vcc, %[q8], %[y8]\n\t"                                                                \
            "s_and_saveexec_b64 %[sv], vcc\n\t"                                                                     \
            "v_fmac_f64_e32 %[a], %[xa], %[vm]\n\t"                                                                 \
            "v_fmac_f64_e32 %[b], %[xb], %[vm]\n\t"                                                                 \
            "s_mov_b64 exec, %[sv]"                                                                                 \
            : [a] "+v"(acc_sub), [b] "+v"(acc_ins), [sv] "=&s"(sv_)                                                 \
            : [y8] "v"(y8v), [xa] "v"(x_sub), [xb] "v"(x_ins), [vm] "v"(vmv), [q8] "n"(8 * (Q))                      \
            : "vcc", "scc");                                                                                               \
    }
#if defined(JTK_PHMM_X_NOBASE)   // timing-only probe: no split by read base
#define BASE_FMAS(xs_, xi_, vmv, y8v) { acc[0] = fma(xs_, vmv, acc[0]); acc[5] = fma(xi_, vmv, acc[5]); }
#elif JTK_PHMM_BASE_CMPX  // v_cmpx writes EXEC itself: 4 instructions per base instead of 5 (round 6: -1.3 % of a pass)
#define BASE_FMAS(xs_, xi_, vmv, y8v)                                                                               \
    {                                                                                                               \
        uint64_t sv_;                                                                                               \
        asm("s_mov_b64 %[sv], exec\n\t"                                                                             \
            "v_cmpx_eq_u32_e32 0, %[y8]\n\t"                                                                        \
            "v_fmac_f64_e32 %[a0], %[xa], %[vm]\n\t"                                                                \
            "v_fmac_f64_e32 %[b0], %[xb], %[vm]\n\t"                                                                \
            "s_mov_b64 exec, %[sv]\n\t"                                                                             \
            "v_cmpx_eq_u32_e32 8, %[y8]\n\t"                                                                        \
            "v_fmac_f64_e32 %[a1], %[xa], %[vm]\n\t"                                                                \
            "v_fmac_f64_e32 %[b1], %[xb], %[vm]\n\t"                                                                \
            "s_mov_b64 exec, %[sv]\n\t"                                                                             \
            "v_cmpx_eq_u32_e32 16, %[y8]\n\t"                                                                       \
            "v_fmac_f64_e32 %[a2], %[xa], %[vm]\n\t"                                                                \
            "v_fmac_f64_e32 %[b2], %[xb], %[vm]\n\t"                                                                \
            "s_mov_b64 exec, %[sv]\n\t"                                                                             \
            "v_cmpx_eq_u32_e32 24, %[y8]\n\t"                                                                       \
            "v_fmac_f64_e32 %[a3], %[xa], %[vm]\n\t"                                                                \
            "v_fmac_f64_e32 %[b3], %[xb], %[vm]\n\t"                                                                \
            "s_mov_b64 exec, %[sv]"                                                                                 \
            : [a0] "+v"(acc[0]), [b0] "+v"(acc[5]), [a1] "+v"(acc[1]), [b1] "+v"(acc[6]), [a2] "+v"(acc[2]),        \
              [b2] "+v"(acc[7]), [a3] "+v"(acc[3]), [b3] "+v"(acc[8]), [sv] "=&s"(sv_)                              \
            : [y8] "v"(y8v), [xa] "v"(xs_), [xb] "v"(xi_), [vm] "v"(vmv)                                            \
            : "vcc", "scc");                                                                                        \
    }
#else
#define BASE_FMAS(xs_, xi_, vmv, y8v)                 \
    BASE_FMA(0, acc[0], acc[5], xs_, xi_, vmv, y8v)   \
    BASE_FMA(1, acc[1], acc[6], xs_, xi_, vmv, y8v)   \
    BASE_FMA(2, acc[2], acc[7], xs_, xi_, vmv, y8v)   \
    BASE_FMA(3, acc[3], acc[8], xs_, xi_, vmv, y8v)
#endif
// acc = fma(y, vd, fma(x, hM, acc)) where the x term only counts on lanes with rowv >= thr_x and the y term on lanes with
// rowv >= thr_y (the del-3 source row i-4 must lie inside the band of ITS diagonal: the one case the three spare lanes of
// the lane ring cannot tell apart)
#define DEL3_FMA(accv, xv, yv, hMv, vdv, rowm4, thr_x, thr_y)                                                       \
    {                                                                                                               \
        uint64_t sv_;                                                                                               \
        asm("v_cmp_le_i32_e32 vcc, %[tx], %[r4]\n\t"                                                                \
            "s_and_saveexec_b64 %[sv], vcc\n\t"                                                                     \
            "v_fmac_f64_e32 %[a], %[x], %[hm]\n\t"                                                                  \
            "s_mov_b64 exec, %[sv]\n\t"                                                                             \
            "v_cmp_le_i32_e32 vcc, %[ty], %[r4]\n\t"                                                                \
            "s_and_saveexec_b64 %[sv], vcc\n\t"                                                                     \
            "v_fmac_f64_e32 %[a], %[y], %[vd]\n\t"                                                                  \
            "s_mov_b64 exec, %[sv]"                                                                                 \
            : [a] "+v"(accv), [sv] "=&s"(sv_)                                                                       \
            : [x] "v"(xv), [y] "v"(yv), [hm] "v"(hMv), [vd] "v"(vdv), [r4] "v"(rowm4), [tx] "s"(thr_x), [ty] "s"(thr_y) \
            : "vcc", "scc");                                                                                               \
    }

#ifndef JTK_PHMM_NUM_VGPR
#define JTK_PHMM_NUM_VGPR 84  // amdgpu_num_vgpr counts the unified file in halves: 168 registers = three waves per SIMD
#endif

// One read: both sweeps.
__device__ __forceinline__ void sweep_read(const int L, const int n, const int r, const int f_lo, const int f_hi, const HmmDev *h,
                                           const uint8_t *gx, const uint8_t *gy, const uint64_t *delta, double2 *scratch,
                                           double *raw, int *rawG, double *lk_out, const uint32_t lds_tmpl,
                                           const uint32_t lds_read) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int T = L + n;
    const uint32_t n_blk = ((lds_tmpl + lds_read) >> 6) + 4;
    const uint32_t S_DELTA = S_VAR, S_EF = S_DELTA + n_blk * 8, S_BFW = S_EF + n_blk * 4,
                   S_XS = (S_BFW + JTK_BFW_BYTES(n_blk) + 15) & ~15u,
                   S_EY = S_XS + ((((lds_tmpl + 2 * PAD + 3) >> 2) + 15) & ~15u);
    // template codes: four 2-bit codes per byte, PAD codes of padding either side (the sweeps look a template row up once per
    // band move and in the generic steps only, so they can afford the unpacking; a byte per code cost a twelfth wave per CU)
    auto xs_of = [&](int i) -> uint32_t {  // 32 * code(x[i-1]): the eM row of template row i, in bytes
        const uint32_t q = (uint32_t)(i - 1 + PAD);
        return ((uint32_t)(*(__attribute__((address_space(3))) const uint8_t *)(uintptr_t)(S_XS + (q >> 2))) >> (2u * (q & 3u)) & 3u) << 5;
    };
    const uint32_t EY0 = S_EY + PAD;      // smem[EY0 + j] = 8 * (y[j-1] | ctx(j) << 2): the eI entry of read column j, in bytes
    uint64_t *s_delta = reinterpret_cast<uint64_t *>(smem + S_DELTA);
    int *s_EF = reinterpret_cast<int *>(smem + S_EF);
    const int lane = threadIdx.x;
    const uint32_t lane16 = lane * 16;
    // ring entry of row i+k for k = -4 .. +2, as byte offsets inside a slot
    uint32_t RK[7];
#pragma unroll
    for (int k = 0; k < 7; k++) RK[k] = (uint32_t)((lane + k - 4) & 63) * 16;
    // The hot LDS accesses address the work area by its byte offset through LDS-address-space pointers (round 6): the kernel has
    // no static __shared__ object, so the dynamic area starts at LDS address 0 (checked below), and an access through `smem + a`
    // costs a `v_add_u32 v, 0, v` per computed address -- the array's base is a link-time constant the compiler cannot fold
    // (40 of them per group of 8 anti-diagonals).
    typedef __attribute__((address_space(3))) const double lds_cf64_t;
    typedef __attribute__((address_space(3))) const uint8_t lds_cu8_t;
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32_t;
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2d_t lds_v2d_t;
    struct RingRef {   // a ring entry: reads and writes as one 16-byte LDS access
        lds_v2d_t *p;
        __device__ __forceinline__ operator double2() const {
            const v2d_t v = *p;
            return make_double2(v.x, v.y);
        }
        __device__ __forceinline__ void operator=(const double2 &v) const {
            v2d_t w;
            w.x = v.x;
            w.y = v.y;
            *p = w;
        }
    };
    auto lds_f64 = [&](uint32_t a) -> double { return *(lds_cf64_t *)(uintptr_t)a; };
    auto lds_u8 = [&](uint32_t a) -> uint32_t { return *(lds_cu8_t *)(uintptr_t)a; };
    auto lds_u32 = [&](uint32_t a) -> uint32_t { return *(lds_cu32_t *)(uintptr_t)a; };
    // (the slot -- a constant in the unrolled groups -- is added as POINTER arithmetic: that is what ends up in the DS instruction's
    // offset field; added to the integer it becomes a v_or_b32 per access)
    typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
    auto ring_at = [&](uint32_t slot, uint32_t off) -> RingRef { return RingRef{(lds_v2d_t *)((lds_byte_t *)(uintptr_t)off + slot * RS)}; };
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();  // (never: see above)
    // four bytes smem[a .. a+3] as one dword (a need not be aligned)
    auto window4 = [&](uint32_t a) -> uint32_t {
        const uint32_t al = a & ~3u;
        return __builtin_amdgcn_alignbyte(lds_u32(al + 4), lds_u32(al), a & 3u);
    };
    auto delta_bit = [&](int t) -> int {  // c[t] - c[t-1], t >= 1 (uniform)
        const uint32_t b = lds_u8(S_DELTA + (uint32_t)(t >> 3));
        return (uni((int)b) >> (t & 7)) & 1;
    };
    auto delta_byte = [&](int k) -> uint32_t { return (uint32_t)uni((int)lds_u8(S_DELTA + (uint32_t)k)); };
    __syncthreads();
    {  // stage the codes (as byte offsets, zero padded), the emission tables and the band deltas
        for (int pb = lane; pb < (L + 2 * PAD + 3) >> 2; pb += 64) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int q = 4 * pb + k - PAD;
                v |= (q >= 0 && q < L) ? (uint32_t)(gx[q] & 3u) << (2 * k) : 0u;
            }
            smem[S_XS + pb] = (uint8_t)v;
        }
                for (int p = lane; p < n + 1 + 2 * PAD + 16; p += 64) {
            const int q = p - PAD;
            smem[S_EY + p] = (q >= 1 && q <= n) ? (uint8_t)(gy[q] << 3) : (uint8_t)0;
        }
        if (lane < 16) reinterpret_cast<double *>(smem + S_EM)[lane] = h->eM[lane];
        if (lane < 20) reinterpret_cast<double *>(smem + S_EI)[lane] = h->eI[lane];
        for (int wdx = lane; wdx < (T >> 6) + 2; wdx += 64) s_delta[wdx] = delta[wdx];
    }
    __syncthreads();
    // scalar operands of every FMA of the recurrences
    const double aMM = uni_f64(h->a[0]), aMI = uni_f64(h->a[1]), aMD = uni_f64(h->a[2]), aIM = uni_f64(h->a[3]),
                 aII = uni_f64(h->a[4]), aID = uni_f64(h->a[5]), aDM = uni_f64(h->a[6]), aDI = uni_f64(h->a[7]),
                 aDD = uni_f64(h->a[8]);
    const uint64_t BAND = (2ull << (2 * r)) - 1;  // 2r+1 ones
#if JTK_PHMM_REPLAY
    // bit g of the replay bits: the backward sweep takes the diagonals 8g .. 8g+7 as a fast group (its own conditions, below) and
    // the forward sweep takes both 8(g-1) .. and 8g .. as fast groups, so that the pairs of the diagonals 8g-5 .. 8g+2 can be
    // replayed from the checkpoint after diagonal 8g-6.  Lane-parallel, once per read.
    for (int gb = 0; gb < (T >> 3) + 10; gb += 64) {
        const int g = gb + lane, tb = 8 * g + 7;
        bool ok = false;
        if (g >= 3 && tb < T && tb <= f_hi && 8 * (g - 1) >= f_lo) {
            const uint32_t b0 = lds_u8(S_DELTA + (uint32_t)g - 1), b1 = lds_u8(S_DELTA + (uint32_t)g), b2 = lds_u8(S_DELTA + (uint32_t)g + 1);
            ok = (b2 & 1u) + (uint32_t)__builtin_popcount((b1 >> 5) & 7u) <= 3u && ((b1 >> 1) & 0xfu) != 0xfu  // backward halves
                 && (b0 & 0xfu) != 0xfu && (b0 >> 4) != 0xfu && (b1 & 0xfu) != 0xfu && (b1 >> 4) != 0xfu;          // forward halves
        }
        const uint64_t w = __ballot(ok);
        if (lane == 0) {
            *reinterpret_cast<uint32_t *>(smem + S_BFW + (uint32_t)(gb >> 3)) = (uint32_t)w;
            *reinterpret_cast<uint32_t *>(smem + S_BFW + (uint32_t)(gb >> 3) + 4) = (uint32_t)(w >> 32);
        }
    }
    __syncthreads();
    auto bf_bits = [&](uint32_t g) -> uint32_t {  // bit 0: group g replays, bit 1: group g+1 (uniform)
        const uint32_t lo = lds_u8(S_BFW + (g >> 3)), hi = lds_u8(S_BFW + (g >> 3) + 1);
        return ((uint32_t)uni((int)(lo | hi << 8)) >> (g & 7u)) & 3u;
    };
#endif

    // =========================== forward ===========================
    int c = 0, EF = 0;                                   // c == c[t-1] between steps
    double toM_1 = 0, toM_2 = 0, toI_1 = 0, toD_1 = 0;  // combos of diagonals t-1 / t-2; exact zeros outside the band
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
    int row = 0;        // fast groups: the lane's template row (a spare lane: the row it takes next)
    uint32_t xrow = 0;  // fast groups: LDS address of the eM row of x[row-1]
    {
        int t = 1;
        bool fast_ready = false;
        uint64_t band = 0;
        int lo6 = 0;
#if JTK_PHMM_FWD_PREFETCH
        uint32_t Wn = 0;           // the read bytes of the coming half group
        double eMn = 0.0, eIn = 0.0;  // the emission entries of the coming step
#endif
        while (t <= T) {
            uint32_t db = 0;
            bool fast = (t & 7) == 0 && t >= f_lo && t + 7 <= f_hi;
#ifdef JTK_PHMM_NOFAST_FWD
#if JTK_PHMM_REPLAY
#error "JTK_PHMM_NOFAST_FWD needs -DJTK_PHMM_REPLAY=0: the replay bits assume the forward sweep's fast groups"
#endif
            fast = false;
#endif
            if (fast) {
                db = delta_byte(t >> 3);
                // a lane that leaves the band takes its next row at the THIRD move after its own (three spare lanes): its
                // registers (row, emission row, read-byte window) are refreshed per half group, so a half may hold 3 moves
                fast = (db & 0xfu) != 0xfu && (db >> 4) != 0xfu;
            }
            if (!fast) {
                // ---- generic step: per-lane predicates (the ends of the sweep)
                fast_ready = false;
                c += delta_bit(t);
                const int lo = c - r, off = (lane - lo) & 63, i = lo + off, j = t - i;
                const bool active = off <= 2 * r && (unsigned)i <= (unsigned)L && (unsigned)j <= (unsigned)n;
                const uint32_t ey8 = lds_u8(EY0 + j), xs = xs_of(i);
                const double eMv = lds_f64(S_EM + xs + (ey8 & 24)), eIv = lds_f64(S_EI + ey8);
                const double pM = rot_from_prev(toM_2), pD = rot_from_prev(toD_1);
                double fm = eMv * pM, fi = eIv * toI_1, fd = pD;
                if (!active) fm = fi = fd = 0.0;
                const double toM_prev = toM_1;
                if ((t & (JTK_SCALE_BLOCK - 1)) == 0) {
                    double mx = fm > fi ? fm : fi;
                    mx = fd > mx ? fd : mx;
                    mx = wave_max(mx);
                    if (mx > 0.0) {
                        const int e = uni(jtk_ilogb_pos(mx));
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
                scratch[(int64_t)t * 64 + lane] = make_double2(toM_prev, toD);
                toM_2 = toM_1;
                toM_1 = toM;
                toI_1 = toI;
                toD_1 = toD;
                if (t == T) {
                    endM = fm;
                    endI = fi;
                    endD = fd;
                }
                t += 1;
                continue;
            }
            // ---- a group of 8 diagonals t .. t+7 inside the interval: EXEC == the band
            MARK("fwd_group_begin");
            if (!fast_ready) {  // c == c[t-1]
                const int lo = c - r;
                lo6 = lo & 63;
                row = lo + ((lane - lo) & 63);
                xrow = S_EM + xs_of(row);
                band = rotl64(BAND, lo6);
                fast_ready = true;
#if JTK_PHMM_FWD_PREFETCH
                Wn = window4(EY0 + (uint32_t)(t - row));
                eMn = lds_f64(xrow | (Wn & 24u));
                eIn = lds_f64(S_EI + (Wn & 0xffu));
#endif
            }
            uint32_t W = 0;  // the read bytes of the half group's columns j .. j+3, j = (t + u) - row
            double2 *out = scratch + (int64_t)t * 64;
#if JTK_PHMM_REPLAY
            // the pairs of the steps 0..2 are the backward group g's, those of the steps 3..7 group g+1's: a group that replays
            // wants the checkpoint after step 2 (in the slots of the diagonals t+3, t+4) instead of its pairs
            const uint32_t bf2 = bf_bits((uint32_t)t >> 3);
#define FWD_STORES(u) ((u) < 3 ? !(bf2 & 1u) : !(bf2 & 2u))
#else
#define FWD_STORES(u) true
#endif
            const bool block_start = (t & (JTK_SCALE_BLOCK - 1)) == 0;  // step u == 0 opens a scaling block
#pragma unroll
            for (int u = 0; u < 8; u++) {
#if JTK_PHMM_FWD_PREFETCH
                // (see the replayed steps in the backward sweep: entries of step u+1 fetched during step u by every lane.  The
                // read bytes of the next half group are loaded at the end of this one -- the same state of `row` as at its top)
                if ((u & 3) == 0) W = Wn;
                const double eMk = eMn, eIk = eIn;
#define FWD_EMISSIONS const double eMv = eMk, eIv = eIk;
#else
                if ((u & 3) == 0) W = window4(EY0 + (uint32_t)(t + u - row));
#define FWD_EMISSIONS const double eMv = lds_f64(xrow | (byte & 24u)), eIv = lds_f64(S_EI + byte);
#endif
                const double pM = rot_from_prev(toM_2), pD = rot_from_prev(toD_1);
                toM_2 = toM_1;  // every lane: a lane outside the band shifts its zeros along
                if ((db >> u) & 1) {  // the band moves up: row c - r leaves it, its lane is spare for the next three moves
#if defined(JTK_PHMM_X_WLANE)
                    toM_1 = wl_zero(toM_1, lo6);
                    toI_1 = wl_zero(toI_1, lo6);
                    toD_1 = wl_zero(toD_1, lo6);
                    row = wlane(c - r + 64, lo6, row);
                    xrow = (uint32_t)wlane((int)(S_EM + (((uint32_t)c & 3u) << 5)), lo6, (int)xrow);
#else
                    if (lanes(1ull << lo6)) {
                        KEEP_MASKED;
                        toM_1 = 0.0;
                        toI_1 = 0.0;
                        toD_1 = 0.0;
                        row += 64;
#ifdef JTK_PHMM_X_NOXS
                        xrow = S_EM + (((uint32_t)row & 3u) << 5);
#else
                        xrow = S_EM + xs_of(row);
#endif
                    }
#endif
                    lo6 = (lo6 + 1) & 63;
                    c += 1;
                    band = (band << 1) | (band >> 63);
                }
                const bool in_band = lanes(band);
#if JTK_PHMM_FWD_PREFETCH
                {
                    if ((u & 3) == 3) Wn = window4(EY0 + (uint32_t)(t + u + 1 - row));
                    const uint32_t bnext = ((u & 3) == 3 ? Wn : W >> (8 * ((u & 3) + 1))) & 0xffu;
#ifdef JTK_PHMM_X_NOEM
                    eMn = jtk_bits_f64(0x3fd0000000000000ull | bnext);
                    eIn = jtk_bits_f64(0x3fd0000000000000ull | xrow);
#else
                    eMn = lds_f64(xrow | (bnext & 24u));
                    eIn = lds_f64(S_EI + bnext);
#endif
                }
#else
                const uint32_t byte = (W >> (8 * (u & 3))) & 0xffu;
#endif
                if (u == 0) {
                    // the group's first diagonal may open a scaling block (the maximum over the band decides the block's
                    // exponent): the band's work is split around that rare step
                    double fm = 0.0, fi = 0.0, fd = 0.0;
                    if (in_band) {
                        KEEP_MASKED;
                        FWD_EMISSIONS
                        fm = eMv * pM;
                        fi = eIv * toI_1;
                        fd = pD;
                    }
                    double sc = 1.0;
                    if (block_start) {
                        KEEP_MASKED;
                        double mx = fm > fi ? fm : fi;
                        mx = fd > mx ? fd : mx;
                        mx = wave_max(mx);
                        if (mx > 0.0) {
                            const int e = uni(jtk_ilogb_pos(mx));
                            sc = pow2i(-e);
                            EF += e;
                        }
                        if (lane == 0) s_EF[t >> 6] = EF;
                        fm *= sc;
                        fi *= sc;
                        fd *= sc;
                    }
                    if (in_band) {
                        KEEP_MASKED;
                        toM_1 = fma(fd, aDM, fma(fi, aIM, fm * aMM));
                        toI_1 = fma(fd, aDI, fma(fi, aII, fm * aMI));
                        toD_1 = fma(fd, aDD, fma(fi, aID, fm * aMD));
                    }
                    if (FWD_STORES(u)) {
                        KEEP_MASKED;
                        out[u * 64 + lane] = make_double2(toM_2, toD_1);  // toM of diagonal t-1 goes out in ITS block's scale ...
                    }
                    if (block_start) {
                        KEEP_MASKED;
                        toM_2 *= sc;  // ... and is re-expressed for the steps that read it
                    }
                } else {
                    if (in_band) {
                        KEEP_MASKED;
                        FWD_EMISSIONS
                        const double fm = eMv * pM, fi = eIv * toI_1, fd = pD;
                        toM_1 = fma(fd, aDM, fma(fi, aIM, fm * aMM));
                        toI_1 = fma(fd, aDI, fma(fi, aII, fm * aMI));
                        toD_1 = fma(fd, aDD, fma(fi, aID, fm * aMD));
                    }
                    if (FWD_STORES(u)) {
                        KEEP_MASKED;
                        out[u * 64 + lane] = make_double2(toM_2, toD_1);
                    }
#if JTK_PHMM_REPLAY
                    if (u == 2 && (bf2 & 2u)) {
                        KEEP_MASKED;
                        out[3 * 64 + lane] = make_double2(toM_1, toM_2);
                        out[4 * 64 + lane] = make_double2(toI_1, toD_1);
                    }
#endif
                }
            }
#undef FWD_STORES
#undef FWD_EMISSIONS
            MARK("fwd_group_end");
            t += 8;
        }
    }
    scratch[(int64_t)(T + 1) * 64 + lane] = make_double2(toM_1, 0.0);  // P_{T+1} = (toM of diagonal T, nothing)
    const int lane_end = L & 63;  // cell (L, n) sits on the lane that owns row L
    double tot = (endM + endI) + endD;
    tot = __shfl(tot, lane_end, 64);
    const double lk = tot > 0.0 ? jtk_log(tot) + (double)EF * JTK_LN2 : JTK_LOG_ZERO;
    if (lane == 0) *lk_out = lk;
    __syncthreads();  // s_EF visible; forward stores are read back by this same wave below

    // =========================== backward + row sums ===========================
    double acc[JTK_ACC_N];
#pragma unroll
    for (int k = 0; k < JTK_ACC_N; k++) acc[k] = 0.0;
    double hM_1 = 0, hM_2 = 0, hI_1 = 0, bD_1 = 0;  // hatM(t+1), hatM(t+2), hatI(t+1), b_D(t+1)
    int EB = 0, Gprev = 0;
    // c == c[T] here (== c[t+1] between steps); c5 = c[t-5], c4 = c[t-4] for the del-3 source rows
    int c5 = c, c4 = c;
    {
        int cc = c, tt = T;
        for (int k = 0; k < 4 && tt >= 1; k++, tt--) cc -= delta_bit(tt);
        c4 = cc;
        if (tt >= 1) cc -= delta_bit(tt);
        c5 = cc;
    }
    auto rel = [&](int d, int blk) -> double {  // 2^(EF[block of diagonal d] - EF[blk])
        if (d < 0 || d > T) return 1.0;
        return fast_pow2(uni(s_EF[d >> 6]) - uni(s_EF[blk]));
    };
    // ring: slot s & 7 holds P_s, in the scale of the block of the step that reads it
    ring_at((T + 2) & 7, lane16) = make_double2(0.0, 0.0);
    for (int ss = T + 1; ss >= T - 4; ss--) {
        double2 v = scratch[(int64_t)ss * 64 + lane];
        v.x *= rel(ss - 1, T >> 6);
        v.y *= rel(ss, T >> 6);
        ring_at(ss & 7, lane16) = v;
    }
    int EFcur = uni(s_EF[T >> 6]);  // forward exponent of the block the sweep is in
    // the row sums of a row: acc[0..3] sub by read base (toM part), acc[4] sub (toD part), acc[5..8] ins, acc[9] ins (toD),
    // acc[10..12] copy 1..3, acc[13..15] del 1..3; vm / vd / hM are the backward values of the cell, a_k the pair of row i+k
#define ROW_SUMS_PLAIN(a_m3, a_m2, a_m1, a_0, a_p1, a_p2, vdv, hMv)                                      \
{                                                                                                  \
    acc[4] = fma((a_m1).y, vdv, acc[4]);                                                           \
    acc[9] = fma((a_0).y, vdv, acc[9]);                                                            \
    acc[10] = fma((a_0).y, vdv, fma((a_0).x, hMv, acc[10]));                                       \
    acc[11] = fma((a_p1).y, vdv, fma((a_p1).x, hMv, acc[11]));                                     \
    acc[12] = fma((a_p2).y, vdv, fma((a_p2).x, hMv, acc[12]));                                     \
    acc[13] = fma((a_m2).y, vdv, fma((a_m2).x, hMv, acc[13]));                                     \
    acc[14] = fma((a_m3).y, vdv, fma((a_m3).x, hMv, acc[14]));                                     \
}
    {
        int t = T;
        bool fast_ready = false;
        uint64_t band = 0;
        int lo6 = 0;
        int rowG = 0;  // exponent of the lane's finished row, until the group's flush
        // pairs on their way from the stripe: qA = those the steps u = 0..3 of the coming group put into the ring (loaded
        // while the previous group ran its steps 4..7), qB = those of the steps u = 4..7 (loaded at the group's start)
#if JTK_PHMM_REPLAY
        // q[k] = the replayed pair of diagonal 8g-5+k (step u of the group puts q[7-u] into the ring); ck0 / ck1 = the
        // checkpoint of the NEXT group below, on its way from the stripe while this group runs
        double2 q[8], ck0 = make_double2(0.0, 0.0), ck1 = ck0;
#else
        double2 qA[4], qB[4];
#pragma unroll
        for (int k = 0; k < 4; k++) qA[k] = qB[k] = make_double2(0.0, 0.0);
#endif
        while (t >= 0) {
            uint32_t w16 = 0;
#if JTK_PHMM_REPLAY
            bool fast = (t & 7) == 7 && (bf_bits((uint32_t)t >> 3) & 1u);
            int carry = 0;
            if (fast) {
                w16 = delta_byte(t >> 3) << 8 | delta_byte((t >> 3) - 1);  // bit k: c[t-15+k] - c[t-16+k]
                carry = delta_bit(t + 1);
            }
#else
            bool fast = (t & 7) == 7 && t - 7 >= f_lo && t <= f_hi && t < T && t >= 23;
            int carry = 0;
            if (fast) {
                w16 = delta_byte(t >> 3) << 8 | delta_byte((t >> 3) - 1);  // bit k: c[t-15+k] - c[t-16+k]
                carry = delta_bit(t + 1);
                // band moves at the steps t .. t-3 / t-4 .. t-7: at most three per half group (see the forward sweep)
                fast = carry + __builtin_popcount((w16 >> 13) & 7u) <= 3 && ((w16 >> 9) & 0xfu) != 0xfu;
            }
#endif
            if (!fast) {
                // ---- generic step
                fast_ready = false;
                if (t < T && (t & 63) == 63) {  // the sweep enters the block below: re-express the ring
                    const int EFabove = EFcur;
                    EFcur = uni(s_EF[t >> 6]);
                    const double f = fast_pow2(EFabove - EFcur);
#pragma unroll
                    for (int sl = 0; sl < 8; sl++) {
                        double2 v = ring_at(sl, lane16);
                        v.x *= f;
                        v.y *= f;
                        ring_at(sl, lane16) = v;
                    }
                }
                const int dn = t < T ? delta_bit(t + 1) : 0;
                c -= dn;  // c == c[t]
                const int lo = c - r, off = (lane - lo) & 63, i = lo + off, j = t - i;
                const bool active = off <= 2 * r && (unsigned)i <= (unsigned)L && (unsigned)j <= (unsigned)n;
                if (dn == 1 && off == 2 * r + 1 && (unsigned)i <= (unsigned)L) {  // the row that left the band is final
                    double2 *dst = reinterpret_cast<double2 *>(raw + (uint64_t)i * JTK_ACC_N);
#pragma unroll
                    for (int k = 0; k < JTK_ACC_N / 2; k++) dst[k] = make_double2(acc[2 * k], acc[2 * k + 1]);
                    rawG[i] = Gprev;
#pragma unroll
                    for (int k = 0; k < JTK_ACC_N; k++) acc[k] = 0.0;
                }
                {  // the pair five diagonals below enters the ring (its slot is not read by this step)
                    double2 v = scratch[(int64_t)(t - 5) * 64 + lane];
                    if ((t & 63) <= 5) {
                        v.x *= rel(t - 6, t >> 6);
                        v.y *= rel(t - 5, t >> 6);
                    }
                    ring_at((t - 5) & 7, lane16) = v;
                }
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
                        const int e = uni(jtk_ilogb_pos(mx));
                        const double sc = pow2i(-e);
                        vm *= sc;
                        vi *= sc;
                        vd *= sc;
                        hM_1 *= sc;
                        EB += e;
                    }
                }
                const uint32_t ey8 = lds_u8(EY0 + j), xs = xs_of(i), y8 = ey8 & 24u;
                const double hM = lds_f64(S_EM + xs + y8) * vm;
                const double hI = lds_f64(S_EI + ey8) * vi;
                const int G = EFcur + EB;
                if (t < T && G != Gprev) {
                    const double sc = pow2i(Gprev - G);
#pragma unroll
                    for (int k = 0; k < JTK_ACC_N; k++) acc[k] *= sc;
                }
                Gprev = G;
                {
                    const uint32_t s0 = (uint32_t)t;
                    double2 a_m4 = ring_at((s0 - 4) & 7, RK[0]);
                    const double2 a_m3 = ring_at((s0 - 3) & 7, RK[1]), a_m2 = ring_at((s0 - 2) & 7, RK[2]),
                                  a_m1 = ring_at((s0 - 1) & 7, RK[3]), a_0 = ring_at(s0 & 7, RK[4]),
                                  a_p1 = ring_at((s0 + 1) & 7, RK[5]), a_p2 = ring_at((s0 + 2) & 7, RK[6]);
                    // the only source row the 3 spare lanes cannot disambiguate
                    if (!(i - 4 >= c5 - r)) a_m4.x = 0.0;
                    if (!(i - 4 >= c4 - r)) a_m4.y = 0.0;
                    double vmq[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) vmq[q] = y8 == 8u * q ? vm : 0.0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        acc[q] = fma(a_m1.x, vmq[q], acc[q]);
                        acc[5 + q] = fma(a_0.x, vmq[q], acc[5 + q]);
                    }
                    ROW_SUMS_PLAIN(a_m3, a_m2, a_m1, a_0, a_p1, a_p2, vd, hM)
                    acc[15] = fma(a_m4.y, vd, fma(a_m4.x, hM, acc[15]));
                }
                hM_2 = hM_1;
                hM_1 = hM;
                hI_1 = hI;
                bD_1 = vd;
                c4 = c5;
                if (t - 5 >= 1) c5 -= delta_bit(t - 5);
                t -= 1;
                continue;
            }
            // ---- a group of 8 diagonals t .. t-7 inside the interval
            MARK("bwd_group_begin");
            if (!fast_ready) {  // c == c[t+1]
                const int lo = c - r, off = (lane - lo) & 63;
                lo6 = lo & 63;
                row = off <= 2 * r ? lo + off : lo + off - 64;
                xrow = S_EM + xs_of(row);
                band = rotl64(BAND, lo6);
                fast_ready = true;
#if JTK_PHMM_REPLAY
                ck0 = scratch[(int64_t)(t - 12) * 64 + lane];
                ck1 = scratch[(int64_t)(t - 11) * 64 + lane];
#else
#pragma unroll
                for (int k = 0; k < 4; k++) qA[k] = scratch[(int64_t)(t - 5 - k) * 64 + lane];
#endif
            }
            const int tb = t;
            uint32_t W = 0;     // read bytes of the half group's columns j-3 .. j, j = (tb - u) - row: step u uses byte 3 - (u & 3)
            uint64_t left = 0;  // lanes whose row left the band in this half group
            const bool low_group = (tb & 63) == 7;  // pairs entering at u >= 2 come from the block below
            const bool block_end = (tb & (JTK_SCALE_BLOCK - 1)) == JTK_SCALE_BLOCK - 1;  // step u == 0 closes a scaling block
            double Fsp = 1.0;
            if (low_group) Fsp = fast_pow2(uni(s_EF[(tb >> 6) - 1]) - uni(s_EF[tb >> 6]));
            const double2 *pin = scratch + (int64_t)(tb - 5) * 64 + lane;  // P_{tb-5}; step u puts pin[-64 u] into the ring
#if JTK_PHMM_REPLAY
            MARK("replay_begin");
            {   // ---- replay the forward steps of the diagonals tb-12 .. tb-5 from the checkpoint after diagonal tb-13
                double fM1 = ck0.x, fM2 = ck0.y, fI1 = ck1.x, fD1 = ck1.y;
                ck0 = pin[-64 * 15];  // the group below, should it replay too: its checkpoint sits in the slots tb-20, tb-19
                ck1 = pin[-64 * 14];
                // c == c[tb+1]; the band of diagonal tb-13 lies carry + (the moves of the diagonals tb-12 .. tb) below it
                const int lof = c - carry - __builtin_popcount(w16 >> 3) - r;
                int lo6f = lof & 63;
                int rowf = lof + ((lane - lof) & 63);
                uint32_t xrowf = S_EM + xs_of(rowf);
                uint32_t yb = EY0 + (uint32_t)(tb - 12 - rowf);  // the read byte of diagonal tb-12+k: smem[yb + k]
                uint64_t bandf = rotl64(BAND, lo6f);
#if JTK_PHMM_REPLAY_PREFETCH
                // The emission entries of step k+1 are fetched during step k, the read byte they are indexed with during step
                // k-1, by EVERY lane: a step otherwise starts with two dependent LDS round trips (byte, then entry) in front of its
                // chain of multiplies.  Valid because a lane's row changes when it LEAVES the band and it is back at the third
                // later move at the earliest: whatever a lane fetched less than three steps ago was fetched with the row it
                // has when it is inside the band (lanes outside fetch something, from the padding at worst, and do not use it).
                uint32_t bn = lds_u8(yb + 1);
                double eMn, eIn;
                {
                    const uint32_t b0 = lds_u8(yb);
                    eMn = lds_f64(xrowf | (b0 & 24u));
                    eIn = lds_f64(S_EI + b0);
                }
#endif
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const double pM = rot_from_prev(fM2), pD = rot_from_prev(fD1);
                    fM2 = fM1;
#if JTK_PHMM_REPLAY_PREFETCH
                    const double eMk = eMn, eIk = eIn;
#define REPLAY_EMISSIONS(kk) const double eMv = eMk, eIv = eIk;
#else
#define REPLAY_EMISSIONS(kk)                    \
    const uint32_t byte = lds_u8(yb + (kk)); \
    const double eMv = lds_f64(xrowf | (byte & 24u)), eIv = lds_f64(S_EI + byte);
#endif
                    if ((w16 >> (3 + k)) & 1u) {
#if defined(JTK_PHMM_X_WLANE)
                        fM1 = wl_zero(fM1, lo6f);
                        fI1 = wl_zero(fI1, lo6f);
                        fD1 = wl_zero(fD1, lo6f);
                        rowf = wlane(lof + k + 64, lo6f, rowf);
                        yb = (uint32_t)wlane((int)(EY0 + (uint32_t)(tb - 12 - lof - k - 64)), lo6f, (int)yb);
                        xrowf = (uint32_t)wlane((int)(S_EM + (((uint32_t)lo6f & 3u) << 5)), lo6f, (int)xrowf);
#else
                        if (lanes(1ull << lo6f)) {
                            KEEP_MASKED;
                            fM1 = 0.0;
                            fI1 = 0.0;
                            fD1 = 0.0;
                            rowf += 64;
                            yb -= 64;
#ifdef JTK_PHMM_X_NOXS
                            xrowf = S_EM + (((uint32_t)rowf & 3u) << 5);
#else
                            xrowf = S_EM + xs_of(rowf);
#endif
                        }
#endif
                        lo6f = (lo6f + 1) & 63;
                        bandf = (bandf << 1) | (bandf >> 63);
                    }
#if JTK_PHMM_REPLAY_PREFETCH
                    if (k < 7) {
#ifdef JTK_PHMM_X_NOEM
                        eMn = jtk_bits_f64(0x3fd0000000000000ull | bn);
                        eIn = jtk_bits_f64(0x3fd0000000000000ull | xrowf);
#else
                        eMn = lds_f64(xrowf | (bn & 24u));
                        eIn = lds_f64(S_EI + bn);
#endif
                        if (k < 6) bn = lds_u8(yb + (uint32_t)(k + 2));
                    }
#endif
                    const bool in_bandf = lanes(bandf);
                    if (k == 5 && low_group) {  // diagonal tb-7 opens a scaling block: the forward step scaled by 2^-e == Fsp
                        KEEP_MASKED;
                        double fm = 0.0, fi = 0.0, fd = 0.0;
                        if (in_bandf) {
                            KEEP_MASKED;
                            REPLAY_EMISSIONS(k)
                            fm = eMv * pM;
                            fi = eIv * fI1;
                            fd = pD;
                        }
                        fm *= Fsp;
                        fi *= Fsp;
                        fd *= Fsp;
                        if (in_bandf) {
                            KEEP_MASKED;
                            fM1 = fma(fd, aDM, fma(fi, aIM, fm * aMM));
                            fI1 = fma(fd, aDI, fma(fi, aII, fm * aMI));
                            fD1 = fma(fd, aDD, fma(fi, aID, fm * aMD));
                        }
                        q[k] = make_double2(fM2, fD1);
                        fM2 *= Fsp;
                    } else {
                        if (in_bandf) {
                            KEEP_MASKED;
                            REPLAY_EMISSIONS(k)
                            const double fm = eMv * pM, fi = eIv * fI1, fd = pD;
                            fM1 = fma(fd, aDM, fma(fi, aIM, fm * aMM));
                            fI1 = fma(fd, aDI, fma(fi, aII, fm * aMI));
                            fD1 = fma(fd, aDD, fma(fi, aID, fm * aMD));
                        }
                        q[k] = make_double2(fM2, fD1);
                    }
#undef REPLAY_EMISSIONS
                }
            }
            MARK("replay_end");
#endif
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if ((u & 3) == 0) W = window4(EY0 + (uint32_t)(tb - u - row - 3));
                if (u == 0 && block_end) {  // the sweep enters the block below: re-express the ring
                    const int EFabove = EFcur;
                    EFcur = uni(s_EF[tb >> 6]);
                    const double f = fast_pow2(EFabove - EFcur);
#pragma unroll
                    for (int sl = 0; sl < 8; sl++) {
                        double2 v = ring_at(sl, lane16);
                        v.x *= f;
                        v.y *= f;
                        ring_at(sl, lane16) = v;
                    }
                }
                const double xm = rot_from_next(hM_2), xd = rot_from_next(bD_1);
                hM_2 = hM_1;  // every lane
                const int dn = u == 0 ? carry : (int)((w16 >> (16 - u)) & 1u);  // c[t+1] - c[t]
                if (dn) {  // the band moves down: row c + r leaves it, final
                    const int hi6 = (lo6 + 2 * r) & 63;
#if defined(JTK_PHMM_X_WLANE)
                    hM_1 = wl_zero(hM_1, hi6);
                    hI_1 = wl_zero(hI_1, hi6);
                    bD_1 = wl_zero(bD_1, hi6);
                    rowG = wlane(Gprev, hi6, rowG);
                    row = wlane(c + r - 64, hi6, row);
                    xrow = (uint32_t)wlane((int)(S_EM + (((uint32_t)c & 3u) << 5)), hi6, (int)xrow);
#else
                    if (lanes(1ull << hi6)) {
                        KEEP_MASKED;
                        hM_1 = 0.0;
                        hI_1 = 0.0;
                        bD_1 = 0.0;
                        rowG = Gprev;
                        row -= 64;
#ifdef JTK_PHMM_X_NOXS
                        xrow = S_EM + (((uint32_t)row & 3u) << 5);
#else
                        xrow = S_EM + xs_of(row);
#endif
                    }
#endif
                    left |= 1ull << hi6;
                    lo6 = (lo6 - 1) & 63;
                    c -= 1;
                    band = (band >> 1) | (band << 63);
                }
#if !JTK_PHMM_REPLAY
                if (u == 0) {
#pragma unroll
                    for (int k = 0; k < 4; k++) qB[k] = pin[-64 * (4 + k)];
                }
#endif
                {  // the pair five diagonals below enters the ring
#if JTK_PHMM_REPLAY
                    double2 v = q[7 - u];
#else
                    double2 v = u < 4 ? qA[u & 3] : qB[u & 3];
                    if (u == 3) {  // qA is free again: the next group's first four pairs (a group that turns out generic
                                   // reloads what it needs)
#pragma unroll
                        for (int k = 0; k < 4; k++) qA[k] = pin[-64 * (8 + k)];
                    }
#endif
                    if (u >= 2 && low_group) {
                        KEEP_MASKED;  // a (rare) uniform branch, not two selects per step
                        v.x *= Fsp;
                        if (u >= 3) v.y *= Fsp;
                    }
                    ring_at((7 - u - 5) & 7, lane16) = v;
                }
                const bool in_band = lanes(band);
                const uint32_t byte = (W >> (8 * (3 - (u & 3)))) & 0xffu;
                const uint32_t y8 = byte & 24u;
                const int thr_x = c5 - r + 4, thr_y = c4 - r + 4;  // the del-3 source row i-4 must be >= c[t-5]-r / c[t-4]-r
#define BWD_PRODUCTS(vmv, vdv, hMv)                                                                                          \
{                                                                                                                        \
    const double2 a_m4 = ring_at((7 - u - 4) & 7, RK[0]);                                                               \
    const double2 a_m3 = ring_at((7 - u - 3) & 7, RK[1]), a_m2 = ring_at((7 - u - 2) & 7, RK[2]),                       \
                  a_m1 = ring_at((7 - u - 1) & 7, RK[3]), a_0 = ring_at((7 - u) & 7, RK[4]),                           \
                  a_p1 = ring_at((7 - u + 1) & 7, RK[5]), a_p2 = ring_at((7 - u + 2) & 7, RK[6]);                       \
    BASE_FMAS(a_m1.x, a_0.x, vmv, y8)                                                                                    \
    ROW_SUMS_PLAIN(a_m3, a_m2, a_m1, a_0, a_p1, a_p2, vdv, hMv)                                                          \
    DEL3_FMA(acc[15], a_m4.x, a_m4.y, hMv, vdv, row, thr_x, thr_y)                                                       \
}
                if (u == 0) {
                    // the group's first diagonal may close a scaling block (the maximum over the band decides the backward
                    // exponent): the band's work is split around that rare step
                    double vm = 0.0, vi = 0.0, vd = 0.0;
                    if (in_band) {
                        KEEP_MASKED;
                        vm = fma(aMD, xd, fma(aMI, hI_1, aMM * xm));
                        vi = fma(aID, xd, fma(aII, hI_1, aIM * xm));
                        vd = fma(aDD, xd, fma(aDI, hI_1, aDM * xm));
                    }
                    if (block_end) {
                        KEEP_MASKED;
                        double mx = vm > vi ? vm : vi;
                        mx = vd > mx ? vd : mx;
                        mx = wave_max(mx);
                        if (mx > 0.0) {
                            const int e = uni(jtk_ilogb_pos(mx));
                            const double sc = pow2i(-e);
                            vm *= sc;
                            vi *= sc;
                            vd *= sc;
                            hM_2 *= sc;  // hatM of diagonal t+1 (already shifted), every lane
                            EB += e;
                        }
                        const int G = EFcur + EB;
                        if (G != Gprev) {
                            const double sca = pow2i(Gprev - G);
                            if (in_band) {
                                KEEP_MASKED;
#pragma unroll
                                for (int k = 0; k < JTK_ACC_N; k++) acc[k] *= sca;
                            }
                        }
                        Gprev = G;
                    }
                    if (in_band) {
                        KEEP_MASKED;
                        const double hM = lds_f64(xrow | y8) * vm, hI = lds_f64(S_EI + byte) * vi;
                        BWD_PRODUCTS(vm, vd, hM)
                        hM_1 = hM;
                        hI_1 = hI;
                        bD_1 = vd;
                    }
                } else {
                    if (in_band) {
                        KEEP_MASKED;
#ifdef JTK_PHMM_X_NOEM
                        const double eMv = jtk_bits_f64(0x3fd0000000000000ull | y8), eIv = jtk_bits_f64(0x3fd0000000000000ull | xrow);
#else
                        const double eMv = lds_f64(xrow | y8), eIv = lds_f64(S_EI + byte);
#endif
                        const double vm = fma(aMD, xd, fma(aMI, hI_1, aMM * xm));
                        const double vi = fma(aID, xd, fma(aII, hI_1, aIM * xm));
                        const double vd = fma(aDD, xd, fma(aDI, hI_1, aDM * xm));
                        const double hM = eMv * vm, hI = eIv * vi;
                        BWD_PRODUCTS(vm, vd, hM)
                        hM_1 = hM;
                        hI_1 = hI;
                        bD_1 = vd;
                    }
                }
#undef BWD_PRODUCTS
                c4 = c5;
                c5 -= (int)((w16 >> (10 - u)) & 1u);  // c[t-5] - c[t-6]
                // the rows that left the band during the half group are final: they sat untouched in their (idle) lanes.  They
                // leave through LDS: a row's 128 bytes go out as ONE line (8 lanes x 16 B) instead of 8 stores of one lane.
                if ((u & 3) == 3 && left != 0) {
                    const uint32_t nleft = (uint32_t)__builtin_popcountll(left);
                    if (lanes(left)) {
                        KEEP_MASKED;
                        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(left >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)left, 0u));
                        double2 *sd = reinterpret_cast<double2 *>(smem + S_STAGE + slot * 128u);
#pragma unroll
                        for (int k = 0; k < JTK_ACC_N / 2; k++) sd[k] = make_double2(acc[2 * k], acc[2 * k + 1]);
                        *reinterpret_cast<int2 *>(smem + S_SMETA + slot * 8u) = make_int2(row + 64, rowG);
#pragma unroll
                        for (int k = 0; k < JTK_ACC_N; k++) acc[k] = 0.0;
                    }
#ifndef JTK_PHMM_X_NOFLUSH
                    if ((uint32_t)lane < 8u * nleft) {
                        KEEP_MASKED;
                        const double2 v = reinterpret_cast<const double2 *>(smem + S_STAGE)[lane];  // slot lane / 8, part lane % 8
                        const int2 meta = *reinterpret_cast<const int2 *>(smem + S_SMETA + (uint32_t)(lane >> 3) * 8u);
                        reinterpret_cast<double2 *>(raw + (uint64_t)meta.x * JTK_ACC_N)[lane & 7] = v;
                        if ((lane & 7) == 0) rawG[meta.x] = meta.y;
                    }
#endif
                }
                if ((u & 3) == 3) left = 0;
            }
            MARK("bwd_group_end");
            t -= 8;
        }
        // rows still in the band after t == 0; c == c[0]
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
#undef ROW_SUMS_PLAIN
}

}  // namespace

__global__ __launch_bounds__(64, 3) __attribute__((amdgpu_num_vgpr(JTK_PHMM_NUM_VGPR))) void phmm_kernel(
    uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks, const ChunkState *state, DevBufs bufs,
    const uint8_t *ey_all, const uint64_t *delta_all, const HmmDev *hmm2, StripeSet stripes,
    uint32_t *work_counter, uint32_t ticket_base, double *raw_all, int *rawG_all, double *lk_all, uint32_t lds_tmpl,
    uint32_t lds_read, int only_active, uint32_t skip_le_radius) {
    const int lane = threadIdx.x;
    // this wave's stripe of the device's forward scratch, for as long as the wave lives (device_common.h: StripeSet);
    // the stripe starts with JTK_SCRATCH_GUARD rows of zeros: "the pair of a diagonal below 0" is an ordinary load
    const uint32_t stripe = jtk_stripe_acquire(stripes);
    double2 *scratch = reinterpret_cast<double2 *>(stripes.mem + (uint64_t)stripe * stripes.stride) + JTK_SCRATCH_GUARD * 64;
#pragma unroll
    for (int g = 1; g <= JTK_SCRATCH_GUARD; g++) scratch[-g * 64 + lane] = make_double2(0.0, 0.0);
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(work_counter, 1u) - ticket_base;  // tickets: the counter is never reset
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_reads) break;
        const ReadMeta rm = reads[item];
        const ChunkMeta cm = chunks[rm.chunk];
        const ChunkState st = state[rm.chunk];
        if (st.status != 0) continue;
        if (only_active && !st.active) continue;
        if (cm.take_num && item - cm.read_first >= cm.take_num) continue;  // this read does not vote
        if (cm.radius > JTK_MAX_RADIUS) continue;                          // phmm_wide_kernel's read
        if (cm.radius <= skip_le_radius) continue;                         // phmm_pair_kernel's read
        const uint64_t *delta = delta_all + rm.delta_off;
        // the interval of diagonals whose whole band lies inside the DP matrix (band_prep_kernel)
        const uint64_t fastw = delta[((cm.tmpl_cap + rm.read_len) >> 6) + 2];
        // everything below was loaded through the vector memory path: make it scalar again (it is the same in every lane),
        // or the sweeps' control flow and addressing would be per-lane
        const int strand = uni((int)rm.strand), buf = uni((int)st.buf);
        sweep_read(uni((int)st.tmpl_len), uni((int)rm.read_len), uni((int)cm.radius), uni((int)(uint32_t)fastw),
                   uni((int)(uint32_t)(fastw >> 32)), hmm2 + (strand ? 0 : 1), bufs.tmpl[buf] + uni64(cm.tmpl_off),
                   ey_all + uni64(rm.ey_off), delta_all + uni64(rm.delta_off), scratch, raw_all + uni64(rm.raw_off),
                   rawG_all + uni64(rm.row_off), lk_all + item, lds_tmpl, lds_read);
    }
    jtk_stripe_release(stripes, stripe);
}

size_t phmm_lds_bytes(uint32_t max_tmpl, uint32_t max_read) {
    const uint32_t n_blk = ((max_tmpl + max_read) >> 6) + 4;
    size_t b = S_VAR + (size_t)n_blk * 12 + JTK_BFW_BYTES(n_blk);
    b = (b + 15) & ~(size_t)15;
    b += ((((max_tmpl + 2 * PAD + 3) >> 2) + 15) & ~15u) + max_read + 1 + 2 * PAD + 16;
    return (b + 15) & ~(size_t)15;
}

void launch_phmm(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                 const ChunkState *state, DevBufs bufs, const uint8_t *ey, const uint64_t *delta,
                 const HmmDev *hmm2, StripeSet stripes, uint32_t n_waves,
                 uint32_t *work_counter, uint32_t *ticket_base, double *raw, int *rawG, double *lk, uint32_t max_tmpl,
                 uint32_t max_read, int only_active, uint32_t skip_le_radius) {
    if (n_reads == 0) return;
    const size_t lds = phmm_lds_bytes(max_tmpl, max_read);
    const uint32_t base = *ticket_base;
    phmm_kernel<<<n_waves, 64, lds, s>>>(n_reads, reads, chunks, state, bufs, ey, delta, hmm2, stripes, work_counter, base, raw,
                                         rawG, lk, max_tmpl, max_read, only_active, skip_le_radius);
    // the host mirror of the never-reset ticket counter moves only when the launch was accepted: a rejected launch (LDS, grid
    // or an earlier sticky error) takes no tickets, and a mirror that ran ahead would make every later launch exit at once
    if (hipPeekAtLastError() == hipSuccess) *ticket_base = base + n_reads + n_waves;
}
