// phmm_pair.hip -- the pair-HMM sweep of phmm_kernels.hip for NARROW bands: two reads of one chunk per wavefront.
//
// With a band radius of 10 (HiFi: ceil(2000 * 0.01) / 2) the 64-lane ring of phmm_kernel keeps 21 lanes busy.  Here a
// wave carries two reads of the same chunk (same template, same radius), one per 32-lane half: lane l owns the template
// row i == (l & 31) (mod 32) inside [c - r, c - r + 31] of ITS read's band, 2r + 1 <= 29 cells + 3 zero lanes per half.
// Everything that is wave-uniform in phmm_kernel (band centre, read length, scaling exponents, band deltas, the strand's
// model) is uniform per HALF here and lives in vector registers; "the value of row i - 1" is a wave rotate by one lane
// with the two lanes at the seam of the halves patched (v_readlane + a select).  Both halves walk the same
// anti-diagonal t at the same step, so the ring slots, the prefetch queue and the unrolling are those of phmm_kernel; a
// read shorter than its partner simply has no cell on the last diagonals.  The arithmetic per cell is phmm_kernel's, bit
// for bit (same specification, same scaling blocks); outputs (raw row sums, exponents, lk) have the same layout, so
// finalize_kernel and everything downstream are shared.
#include <type_traits>

#include "device_common.h"

namespace {

#define PR_PAD 64   // padding (bytes) around the staged code arrays
#define PR_RW 38    // ring entries per half of a slot: 32 lanes + 4 wrapped copies in front + 2 behind
#define PR_SLOT (2 * PR_RW)
#define PR_PF 4     // pairs in flight from HBM during the backward sweep (the unrolling assumes 4)

__device__ __forceinline__ int rd_lane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
// lane l <- lane ((l & 31) - 1) & 31 of the same half
__device__ __forceinline__ double rot32_from_prev(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int rlo = __builtin_amdgcn_update_dpp(0, lo, 0x13C, 0xF, 0xF, false);  // wave_ror:1: lane l <- lane (l - 1) & 63
    int rhi = __builtin_amdgcn_update_dpp(0, hi, 0x13C, 0xF, 0xF, false);
    // the seam: lane 0 must take lane 31 (it got 63), lane 32 must take lane 63 (it got 31)
    const int a_lo = rd_lane(lo, 31), a_hi = rd_lane(hi, 31), b_lo = rd_lane(lo, 63), b_hi = rd_lane(hi, 63);
    const int lane = (int)threadIdx.x;
    if ((lane & 31) == 0) {
        rlo = lane ? b_lo : a_lo;
        rhi = lane ? b_hi : a_hi;
    }
    return __hiloint2double(rhi, rlo);
}
// lane l <- lane ((l & 31) + 1) & 31 of the same half
__device__ __forceinline__ double rot32_from_next(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int rlo = __builtin_amdgcn_update_dpp(0, lo, 0x134, 0xF, 0xF, false);  // wave_rol:1: lane l <- lane (l + 1) & 63
    int rhi = __builtin_amdgcn_update_dpp(0, hi, 0x134, 0xF, 0xF, false);
    // the seam: lane 31 must take lane 0 (it got 32), lane 63 must take lane 32 (it got 0)
    const int a_lo = rd_lane(lo, 0), a_hi = rd_lane(hi, 0), b_lo = rd_lane(lo, 32), b_hi = rd_lane(hi, 32);
    const int lane = (int)threadIdx.x;
    if ((lane & 31) == 31) {
        rlo = lane == 63 ? b_lo : a_lo;
        rhi = lane == 63 ? b_hi : a_hi;
    }
    return __hiloint2double(rhi, rlo);
}
__device__ __forceinline__ double half_max(double v) {  // maximum over the lane's 32-lane half
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        double u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}
__device__ __forceinline__ double pr_pow2(int e) {  // 2^e, lane-varying exponent
    if (e > -1000 && e < 1000) return jtk_bits_f64((uint64_t)(1023 + e) << 52);
    return jtk_scalbn(1.0, e);
}

__global__ __launch_bounds__(64, 2) void phmm_pair_kernel(uint32_t n_items, const uint32_t *items, const ReadMeta *reads,
                                                         const ChunkMeta *chunks, const ChunkState *state, DevBufs bufs,
                                                         const uint8_t *ey_all, const uint64_t *delta_all, const HmmDev *hmm2,
                                                         StripeSet stripes, uint32_t *work_counter, uint32_t ticket_base,
                                                         double *raw_all, int *rawG_all, double *lk_all, uint32_t lds_tmpl,
                                                         uint32_t lds_read, int only_active) {
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS carve: ring [8][2][PR_RW] double2 | per half: eM[16] eI[20] | per half: delta words | per half: block exponents |
    // template codes (shared) | per half: read codes
    double2 *ring = reinterpret_cast<double2 *>(smem);
    unsigned char *s_tab = smem + 8 * PR_SLOT * 16;  // 2 x (16 + 20) doubles
    const uint32_t n_blk = ((lds_tmpl + lds_read) >> 6) + 4;
    uint64_t *s_delta = reinterpret_cast<uint64_t *>(s_tab + 2 * 36 * 8);  // [2][n_blk]
    int *s_EF = reinterpret_cast<int *>(s_delta + 2 * n_blk);               // [2][n_blk]
    uint8_t *s_xs = reinterpret_cast<uint8_t *>(s_EF + 2 * n_blk);
    const uint32_t xs_bytes = (lds_tmpl + 2 * PR_PAD + 15) & ~15u, ey_bytes = (lds_read + 1 + 2 * PR_PAD + 15) & ~15u;
    uint8_t *s_ey = s_xs + xs_bytes;  // [2][ey_bytes]
    const int lane = threadIdx.x, hb = lane >> 5, l32 = lane & 31;
    // the stripe starts with JTK_SCRATCH_GUARD rows of zeros: "the pair of a diagonal below 0" is then an ordinary load
    const uint32_t stripe = jtk_stripe_acquire(stripes);  // for as long as the wave lives (device_common.h: StripeSet)
    double2 *scratch = reinterpret_cast<double2 *>(stripes.mem + (uint64_t)stripe * stripes.stride) + JTK_SCRATCH_GUARD * 64;
#pragma unroll
    for (int g = 1; g <= JTK_SCRATCH_GUARD; g++) scratch[-g * 64 + lane] = make_double2(0.0, 0.0);
    // this half's views
    const unsigned char *t_eM = s_tab + hb * 36 * 8, *t_eI = t_eM + 16 * 8;
    uint64_t *h_delta = s_delta + hb * n_blk;
    int *h_EF = s_EF + hb * n_blk;
    const uint8_t *xs0 = s_xs + PR_PAD - 1;                 // xs0[i] = row offset of x[i-1]
    const uint8_t *ey0 = s_ey + hb * ey_bytes + PR_PAD;     // ey0[j] = entry offset of this half's read base j

    for (;;) {
        uint32_t q = 0;
        if (lane == 0) q = atomicAdd(work_counter, 1u) - ticket_base;  // tickets: the counter is never reset
        q = __builtin_amdgcn_readfirstlane(q);
        if (q >= n_items) break;
        const uint32_t it = items[q], first = it & 0x7fffffffu;
        const bool single = (it >> 31) != 0;
        const ReadMeta rmA = reads[first];
        const ChunkMeta cm = chunks[rmA.chunk];
        const ChunkState st = state[rmA.chunk];
        if (st.status != 0) continue;
        if (only_active && !st.active) continue;
        const ReadMeta rmB = reads[single ? first : first + 1];
        const bool valid = hb == 0 || !single;  // the second half of a single item idles
        const int L = (int)st.tmpl_len, r = (int)cm.radius;
        const int n = valid ? (int)(hb ? rmB.read_len : rmA.read_len) : 0;
        const int T = L + n;                                      // this half's last diagonal
        const int nA = (int)rmA.read_len, nB = single ? 0 : (int)rmB.read_len;
        const int Tmax = L + (nA > nB ? nA : nB);                 // wave-uniform
        const uint32_t my_item = hb ? first + 1 : first;
        const uint32_t my_strand = hb ? rmB.strand : rmA.strand;
        const HmmDev *h = hmm2 + (my_strand ? 0 : 1);
        __syncthreads();
        // ---- stage: template codes once, the two reads' codes, both strands' tables, both reads' band deltas
        {
            const uint8_t *gx = bufs.tmpl[st.buf] + cm.tmpl_off;
            for (int p = lane; p < L + 2 * PR_PAD; p += 64) {
                const int qq = p - PR_PAD;
                s_xs[p] = (qq >= 0 && qq < L) ? (uint8_t)(gx[qq] << 5) : (uint8_t)0;
            }
            for (int hh = 0; hh < 2; hh++) {
                const ReadMeta &rm = hh ? rmB : rmA;
                const int nn = (hh && single) ? 0 : (int)rm.read_len;
                const uint8_t *gy = ey_all + rm.ey_off;
                uint8_t *dst = s_ey + hh * ey_bytes;
                for (int p = lane; p < nn + 1 + 2 * PR_PAD; p += 64) {
                    const int qq = p - PR_PAD;
                    dst[p] = (qq >= 1 && qq <= nn) ? (uint8_t)(gy[qq] << 3) : (uint8_t)0;
                }
                const HmmDev *hh_m = hmm2 + (rm.strand ? 0 : 1);
                double *tab = reinterpret_cast<double *>(s_tab + hh * 36 * 8);
                if (lane < 16) tab[lane] = hh_m->eM[lane];
                if (lane < 20) tab[16 + lane] = hh_m->eI[lane];
                const uint64_t *gd = delta_all + rm.delta_off;
                const int words = ((L + nn) >> 6) + 2;
                for (int w = lane; w < (int)n_blk; w += 64) s_delta[hh * n_blk + w] = (w < words && !(hh && single)) ? gd[w] : 0ull;
            }
        }
        __syncthreads();
        const double aMM = h->a[0], aMI = h->a[1], aMD = h->a[2], aIM = h->a[3], aII = h->a[4], aID = h->a[5], aDM = h->a[6],
                     aDI = h->a[7], aDD = h->a[8];
        auto dbit = [&](int t) -> int {  // c[t] - c[t-1] of this half's read (0 beyond its last diagonal: the words are zero)
            return (int)((h_delta[t >> 6] >> (t & 63)) & 1ull);
        };

        // =========================== forward ===========================
        int c = 0, EF = 0;
        double toM_1 = 0, toM_2 = 0, toI_1 = 0, toD_1 = 0;
        double endM = 0, endI = 0, endD = 0;
        {
            const double fm = (l32 == 0 && valid) ? 1.0 : 0.0;
            toM_1 = fm * aMM;
            toI_1 = fm * aMI;
            toD_1 = fm * aMD;
            scratch[lane] = make_double2(0.0, toD_1);
            if (l32 == 0) h_EF[0] = 0;
            if (T == 0) endM = fm;
        }
        uint64_t dw = h_delta[0];
        int c_n = 0;
        bool act_n = false;
        double eM_n = 0.0, eI_n = 0.0;
        auto prefetch_fwd = [&](int tn) {
            if ((tn & 63) == 0) dw = h_delta[tn >> 6];
            c_n += (int)((dw >> (tn & 63)) & 1ull);
            const int lo = c_n - r, off = (l32 - lo) & 31, i = lo + off, j = tn - i;
            act_n = valid && off <= 2 * r && (unsigned)i <= (unsigned)L && (unsigned)j <= (unsigned)n;
            const int jj = (unsigned)(j + PR_PAD) <= (unsigned)(lds_read + 2 * PR_PAD) ? j : 0;  // the partner may be longer
            const int ey8 = ey0[jj], xs = xs0[i];
            eM_n = *reinterpret_cast<const double *>(t_eM + xs + (ey8 & 24));
            eI_n = *reinterpret_cast<const double *>(t_eI + ey8);
        };
        if (Tmax >= 1) prefetch_fwd(1);
        for (int t = 1; t <= Tmax; t++) {
            if (t <= T) c = c_n;
            const bool active = act_n;
            const double eMv = eM_n, eIv = eI_n;
            if (t < Tmax) prefetch_fwd(t + 1);
            const double pM = rot32_from_prev(toM_2), pD = rot32_from_prev(toD_1);
            double fm = eMv * pM, fi = eIv * toI_1, fd = pD;
            if (!active) fm = fi = fd = 0.0;
            const double toM_prev = toM_1;
            if ((t & (JTK_SCALE_BLOCK - 1)) == 0) {
                double mx = fm > fi ? fm : fi;
                mx = fd > mx ? fd : mx;
                mx = half_max(mx);
                if (mx > 0.0) {
                    const int e = jtk_ilogb_pos(mx);
                    const double sc = pr_pow2(-e);
                    fm *= sc;
                    fi *= sc;
                    fd *= sc;
                    toM_1 *= sc;
                    EF += e;
                }
                if (l32 == 0) h_EF[t >> 6] = EF;
            }
            const double toM = fma(fd, aDM, fma(fi, aIM, fm * aMM));
            const double toI = fma(fd, aDI, fma(fi, aII, fm * aMI));
            const double toD = fma(fd, aDD, fma(fi, aID, fm * aMD));
            // the pair's first entry is toM of diagonal t-1 in its own block's scale (the value before this step's rescale)
#if defined(JTK_PAIR_X_NOSTREAM)  // timing only (scripts/pair_probe_r6.sh): no pair leaves the wave but the ones the sweep back starts from
            if (t >= Tmax - 5) scratch[(uint64_t)t * 64 + lane] = make_double2(toM_prev, toD);
#elif defined(JTK_PAIR_X_REPLAYCOST)  // timing only: one row per four diagonals (a checkpoint would be two rows per eight)
            if (t >= Tmax - 5 || (t & 3) == 2) scratch[(uint64_t)t * 64 + lane] = make_double2(toM_prev, toD);
#else
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
        // P_{Tmax+1} = (toM of diagonal Tmax, nothing); for the shorter read the loop itself wrote its P_{T+1}
        scratch[(uint64_t)(Tmax + 1) * 64 + lane] = make_double2(toM_1, 0.0);
        {
            const int lane_end = hb * 32 + (L & 31);
            double tot = (endM + endI) + endD;
            tot = __shfl(tot, lane_end, 64);
            const double lk = tot > 0.0 ? jtk_log(tot) + (double)EF * JTK_LN2 : JTK_LOG_ZERO;
            if (l32 == 0 && valid) lk_all[my_item] = lk;
        }
        __syncthreads();

        // =========================== backward + table accumulation ===========================
        const ReadMeta &rmMine = hb ? rmB : rmA;
        double *raw = raw_all + rmMine.raw_off;
        int *rawG = rawG_all + rmMine.row_off;
        double acc[JTK_ACC_N];
#pragma unroll
        for (int k = 0; k < JTK_ACC_N; k++) acc[k] = 0.0;
        double hM_1 = 0, hM_2 = 0, hI_1 = 0, bD_1 = 0;
        int EB = 0, Gprev = 0;
        int c5 = c, c4 = c;  // c == c[T] of this half
        {
            int cc = c, tt = T;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (tt >= 1) {
                    cc -= dbit(tt);
                    tt--;
                }
            }
            c4 = cc;
            if (tt >= 1) cc -= dbit(tt);
            c5 = cc;
        }
        auto load_pair = [&](int ss) -> double2 { return scratch[(int64_t)ss * 64 + lane]; };  // ss >= -JTK_SCRATCH_GUARD
        double2 *ring_me = ring + hb * PR_RW + l32 + 4;  // this lane's entry of slot 0
        const int wrap_off = l32 < 2 ? 32 : (l32 >= 28 ? -32 : 0);  // the wrapped copy, or the lane's own entry again
        auto ring_put_at = [&](double2 *e, double2 v) __attribute__((always_inline)) {
            e[0] = v;
            e[wrap_off] = v;
        };
        auto ring_put = [&](int slot, double2 v) __attribute__((always_inline)) { ring_put_at(ring_me + slot * PR_SLOT, v); };
        auto rel = [&](int d, int blk) -> double {  // 2^(EF[block of diagonal d] - EF[blk]) of this half
            if (d < 0 || d > T) return 1.0;
            return pr_pow2(h_EF[d >> 6] - h_EF[blk]);
        };
        ring_put((Tmax + 2) & 7, make_double2(0.0, 0.0));
        for (int ss = Tmax + 1; ss >= Tmax - 4; ss--) {
            double2 v = load_pair(ss);
            v.x *= rel(ss - 1, Tmax >> 6);
            v.y *= rel(ss, Tmax >> 6);
            ring_put(ss & 7, v);
        }
        // one batch of loads per group of four diagonals, a group ahead (see phmm_kernel)
        auto s_of = [&](int tb, int idx) -> int { return tb - 5 - ((2 - idx) & 3); };
        double2 pq[PR_PF], pqY[PR_PF];
#pragma unroll
        for (int qq = 0; qq < PR_PF; qq++) pq[qq] = load_pair(s_of(Tmax | 3, qq));
        int delta_next = 0;
        int EFcur = h_EF[Tmax >> 6];
        double2 *half[2] = {ring_me, ring_me + 4 * PR_SLOT};
        auto step = [&](int t, auto u_tag, auto pq_tag) __attribute__((always_inline)) {
            constexpr int pq_idx = decltype(pq_tag)::value;
            constexpr int U = decltype(u_tag)::value;
            auto entry = [&](int dx) -> double2 * {
                const int lo2 = (3 - U) + dx;
                return half[(lo2 >> 2) & 1] + (lo2 & 3) * PR_SLOT;
            };
            if (U == 0 && (t & 63) == 63 && t < Tmax) {  // the sweep enters the block below
                const int EFabove = EFcur;
                EFcur = h_EF[t >> 6];
                const double f = pr_pow2(EFabove - EFcur);
#pragma unroll
                for (int sl = 0; sl < 8; sl++) {
                    double2 v = ring_me[sl * PR_SLOT];
                    v.x *= f;
                    v.y *= f;
                    ring_put(sl, v);
                }
            }
            if (t < T) c -= delta_next;
            const int lo = c - r, off = (l32 - lo) & 31, i = lo + off, j = t - i;
            const bool active = valid && off <= 2 * r && (unsigned)i <= (unsigned)L && (unsigned)j <= (unsigned)n;
            if (valid && t < T && delta_next == 1) {
                if (off == 2 * r + 1 && (unsigned)i <= (unsigned)L) {
                    double2 *dst = reinterpret_cast<double2 *>(raw + (uint64_t)i * JTK_ACC_N);
#pragma unroll
                    for (int k = 0; k < JTK_ACC_N / 2; k++) dst[k] = make_double2(acc[2 * k], acc[2 * k + 1]);
                    rawG[i] = Gprev;
#pragma unroll
                    for (int k = 0; k < JTK_ACC_N; k++) acc[k] = 0.0;
                }
            }
            {
                double2 v = pq[pq_idx];
                if ((t & 63) <= 5) {
                    v.x *= rel(t - 6, t >> 6);
                    v.y *= rel(t - 5, t >> 6);
                }
                ring_put_at(entry(-5), v);
            }
            double vm, vi, vd;
            {
                const double xm = rot32_from_next(hM_2), xd = rot32_from_next(bD_1), xi = hI_1;
                vm = fma(aMD, xd, fma(aMI, xi, aMM * xm));
                vi = fma(aID, xd, fma(aII, xi, aIM * xm));
                vd = fma(aDD, xd, fma(aDI, xi, aDM * xm));
                if (t == T) vm = vi = vd = 1.0;
            }
            if (!active) vm = vi = vd = 0.0;
            if ((t & (JTK_SCALE_BLOCK - 1)) == JTK_SCALE_BLOCK - 1 && t < Tmax) {
                double mx = vm > vi ? vm : vi;
                mx = vd > mx ? vd : mx;
                mx = half_max(mx);
                if (t < T && mx > 0.0) {
                    const int e = jtk_ilogb_pos(mx);
                    const double sc = pr_pow2(-e);
                    vm *= sc;
                    vi *= sc;
                    vd *= sc;
                    hM_1 *= sc;
                    EB += e;
                }
            }
            const int jj = (unsigned)(j + PR_PAD) <= (unsigned)(lds_read + 2 * PR_PAD) ? j : 0;
            const int ey8 = ey0[jj], xs = xs0[i], y8 = ey8 & 24;
            const double hM = *reinterpret_cast<const double *>(t_eM + xs + y8) * vm;
            const double hI = *reinterpret_cast<const double *>(t_eI + ey8) * vi;
            const int G = EFcur + EB;
            {
                const bool need = t < T && G != Gprev;
                if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                    const double sc = need ? pr_pow2(Gprev - G) : 1.0;
#pragma unroll
                    for (int k = 0; k < JTK_ACC_N; k++) acc[k] *= sc;
                }
            }
            Gprev = G;
            auto pair = [&](int x, int k) -> double2 { return entry(x)[k]; };
            double vmq[4];
#pragma unroll
            for (int qq = 0; qq < 4; qq++) vmq[qq] = y8 == 8 * qq ? vm : 0.0;
            {
                const double2 a = pair(-1, -1);
#pragma unroll
                for (int qq = 0; qq < 4; qq++) acc[qq] = fma(a.x, vmq[qq], acc[qq]);
                acc[4] = fma(a.y, vd, acc[4]);
            }
            {
                const double2 a = pair(0, 0);
#pragma unroll
                for (int qq = 0; qq < 4; qq++) acc[5 + qq] = fma(a.x, vmq[qq], acc[5 + qq]);
                acc[9] = fma(a.y, vd, acc[9]);
                acc[10] = fma(a.y, vd, fma(a.x, hM, acc[10]));
            }
#pragma unroll
            for (int cc = 2; cc <= 3; cc++) {
                const double2 a = pair(cc - 1, cc - 1);
                acc[10 + cc - 1] = fma(a.y, vd, fma(a.x, hM, acc[10 + cc - 1]));
            }
#pragma unroll
            for (int dd = 1; dd <= 3; dd++) {
                double2 a = pair(-dd - 1, -dd - 1);
                if (dd == 3) {
                    if (!(i - 4 >= c5 - r)) a.x = 0.0;
                    if (!(i - 4 >= c4 - r)) a.y = 0.0;
                }
                acc[13 + dd - 1] = fma(a.y, vd, fma(a.x, hM, acc[13 + dd - 1]));
            }
            hM_2 = hM_1;
            hM_1 = hM;
            hI_1 = hI;
            bD_1 = vd;
            if (t <= T) {  // the shorter read's centres wait at c[T] until the sweep reaches its last diagonal
                delta_next = t >= 1 ? dbit(t) : 0;
                c4 = c5;
                if (t - 5 >= 1) c5 -= dbit(t - 5);
            }
        };
        for (int tb = Tmax | 3; tb >= 3; tb -= 4) {
            {
                double2 *lo_half = ring_me, *hi_half = lo_half + 4 * PR_SLOT;
                const bool hi = (tb >> 2) & 1;
                half[0] = hi ? hi_half : lo_half;
                half[1] = hi ? lo_half : hi_half;
            }
#define PR_GROUP_STEP(u)                                                                                     \
    {                                                                                                        \
        const int t = tb - (u);                                                                              \
        if (t <= Tmax) step(t, std::integral_constant<int, (u)>{}, std::integral_constant<int, (2 - (u)) & 3>{}); \
    }
            PR_GROUP_STEP(0)
#if defined(JTK_PAIR_X_NOSTREAM)
            // timing only: the sweep back without its stream of pairs (the bound on what any replay can win); the tables are garbage
#pragma unroll
            for (int qq = 0; qq < PR_PF; qq++) pqY[qq] = pq[qq];
#elif defined(JTK_PAIR_X_REPLAYCOST)
            // timing only: what a replay would cost -- one row loaded per group of four (the checkpoint's share), then the forward
            // step of four diagonals (band position, emission look-ups, the two seam-patched rotates, the nine products, the block's
            // rescale from the stored exponents) whose pairs are what the next group consumes; the state is seeded from the
            // loaded row, so the values are of the right magnitude but the tables are garbage
            {
                const double2 seed = load_pair(tb - 9);  // row == 2 (mod 4): stored by the forward sweep of this build
                double rM1 = seed.x, rM2 = seed.x, rI1 = seed.y, rD1 = seed.y;
                int rc = c5;
#pragma unroll
                for (int qq = 0; qq < PR_PF; qq++) {
                    const int tn = tb - 12 + qq;
                    if (tn >= 1) rc += dbit(tn) - dbit(tn + 4);  // a centre that moves like the band's
                    const int lo = rc - r, off = (l32 - lo) & 31, i = lo + off, j = tn - i;
                    const bool act = valid && off <= 2 * r && (unsigned)i <= (unsigned)L && (unsigned)j <= (unsigned)n;
                    const int jj = (unsigned)(j + PR_PAD) <= (unsigned)(lds_read + 2 * PR_PAD) ? j : 0;
                    const int ii = (unsigned)(i + PR_PAD - 1) <= (unsigned)(lds_tmpl + 2 * PR_PAD - 2) ? i : 0;
                    const int ey8 = ey0[jj], xs = xs0[ii];
                    const double eMv = *reinterpret_cast<const double *>(t_eM + xs + (ey8 & 24));
                    const double eIv = *reinterpret_cast<const double *>(t_eI + ey8);
                    const double pM = rot32_from_prev(rM2), pD = rot32_from_prev(rD1);
                    double fm = eMv * pM, fi = eIv * rI1, fd = pD;
                    if (!act) fm = fi = fd = 0.0;
                    const double rM_prev = rM1;
                    if ((tn & (JTK_SCALE_BLOCK - 1)) == 0 && tn >= JTK_SCALE_BLOCK) {
                        const double sc = pr_pow2(h_EF[(tn >> 6) - 1] - h_EF[tn >> 6]);
                        fm *= sc;
                        fi *= sc;
                        fd *= sc;
                        rM1 *= sc;
                    }
                    const double toM = fma(fd, aDM, fma(fi, aIM, fm * aMM));
                    const double toI = fma(fd, aDI, fma(fi, aII, fm * aMI));
                    const double toD = fma(fd, aDD, fma(fi, aID, fm * aMD));
                    pqY[(qq + 3) & 3] = make_double2(rM_prev, toD);
                    rM2 = rM1;
                    rM1 = toM;
                    rI1 = toI;
                    rD1 = toD;
                }
            }
#else
#pragma unroll
            for (int qq = 0; qq < PR_PF; qq++) pqY[qq] = load_pair(s_of(tb, qq) - 4);
#endif
            PR_GROUP_STEP(1)
            PR_GROUP_STEP(2)
            PR_GROUP_STEP(3)
#undef PR_GROUP_STEP
#pragma unroll
            for (int qq = 0; qq < PR_PF; qq++) pq[qq] = pqY[qq];
        }
        {
            const int lo = c - r, off = (l32 - lo) & 31, i = lo + off;
            if (valid && off <= 2 * r && (unsigned)i <= (unsigned)L) {
                double2 *dst = reinterpret_cast<double2 *>(raw + (uint64_t)i * JTK_ACC_N);
#pragma unroll
                for (int k = 0; k < JTK_ACC_N / 2; k++) dst[k] = make_double2(acc[2 * k], acc[2 * k + 1]);
                rawG[i] = Gprev;
            }
        }
    }
    jtk_stripe_release(stripes, stripe);
}

}  // namespace

size_t phmm_pair_lds_bytes(uint32_t max_tmpl, uint32_t max_read) {
    const uint32_t n_blk = ((max_tmpl + max_read) >> 6) + 4;
    size_t b = 8 * PR_SLOT * 16 + 2 * 36 * 8 + (size_t)2 * n_blk * 12;
    b += ((max_tmpl + 2 * PR_PAD + 15) & ~15u) + 2 * (size_t)((max_read + 1 + 2 * PR_PAD + 15) & ~15u);
    return (b + 15) & ~(size_t)15;
}

void launch_phmm_pair(hipStream_t s, uint32_t n_items, const uint32_t *items, const ReadMeta *reads, const ChunkMeta *chunks,
                      const ChunkState *state, DevBufs bufs, const uint8_t *ey, const uint64_t *delta, const HmmDev *hmm2,
                      StripeSet stripes, uint32_t n_waves, uint32_t *work_counter, uint32_t *ticket_base, double *raw, int *rawG,
                      double *lk, uint32_t max_tmpl, uint32_t max_read, int only_active) {
    if (n_items == 0) return;
    const uint32_t base = *ticket_base;
    const size_t lds = phmm_pair_lds_bytes(max_tmpl, max_read);
    phmm_pair_kernel<<<n_waves, 64, lds, s>>>(n_items, items, reads, chunks, state, bufs, ey, delta, hmm2, stripes,
                                              work_counter, base, raw, rawG, lk, max_tmpl, max_read, only_active);
    // the host mirror of the never-reset ticket counter moves only when the launch was accepted: a rejected launch (LDS, grid
    // or an earlier sticky error) takes no tickets, and a mirror that ran ahead would make every later launch exit at once
    if (hipPeekAtLastError() == hipSuccess) *ticket_base = base + n_items + n_waves;
}
