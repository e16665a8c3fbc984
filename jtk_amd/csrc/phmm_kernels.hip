// phmm_kernels.hip -- banded pair-HMM forward/backward + modification table for gfx950 (CDNA4).
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
//  * the forward sweep streams (toM, toD) of every anti-diagonal to a per-wave scratch stripe in HBM as
//    one coalesced 1 KiB store (16 B per lane); the backward sweep reads them back through a register
//    prefetch queue into an 8-slot LDS ring, from which the 16 row-crossing products of a step are read
//    with the lane offset folded into the LDS address;
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
    for (uint32_t k = 0; k < n_ops; k++) {
        uint8_t op = ops[k];
        if (op == JTK_OP_INS) {
            t += 1;
            j++;
        } else if (op == JTK_OP_DEL) {
            t += 1;
            if (t <= T) d[t >> 6] |= 1ull << (t & 63);
            i++;
        } else if (op <= JTK_OP_MISMATCH) {
            t += 1;
            if (t <= T) d[t >> 6] |= 1ull << (t & 63);
            t += 1;
            i++;
            j++;
        } else {
            bad = true;
        }
        if (t > T) {
            bad = true;
            break;
        }
    }
    if (bad || i != L || j != n) atomicMin(&st->status, (int)JTK_ERR_OPS_MISMATCH);
}

// ------------------------------------------------------------------------------------------------------
// The per-read forward/backward sweep.
// ------------------------------------------------------------------------------------------------------
#define PF 4  // prefetch queue depth of the backward sweep (register staged, compiler-managed vmcnt)

struct Lane {
    int i, j;
    bool active;
};
__device__ __forceinline__ Lane lane_cell(int lane, int c, int r, int t, int L, int n) {
    Lane x;
    const int lo = c - r;
    const int off = (lane - lo) & 63;
    x.i = lo + off;
    x.j = t - x.i;
    x.active = off <= 2 * r && x.i >= 0 && x.i <= L && x.j >= 0 && x.j <= n;
    return x;
}

__global__ __launch_bounds__(64, 3) void phmm_kernel(uint32_t n_reads, const ReadMeta *reads,
                                                  const ChunkMeta *chunks, const ChunkState *state,
                                                  DevBufs bufs, const uint8_t *ey_all, const uint64_t *delta_all,
                                                  const HmmDev *hmm2, double *scratch_all,
                                                  uint64_t scratch_stride, uint32_t *work_counter, double *raw_all,
                                                  int *rawG_all, double *lk_all, uint32_t lds_tmpl,
                                                  uint32_t lds_read, int only_active) {
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS carve: ring [8][64] double2 | eM[16] | eI[20] | expo[ (Tmax>>6)+2 ] int | tmpl codes | ey
    double2 *ring = reinterpret_cast<double2 *>(smem);
    double *s_eM = reinterpret_cast<double *>(smem + 8 * 64 * 16);
    double *s_eI = s_eM + 16;
    const uint32_t n_blk = ((lds_tmpl + lds_read) >> 6) + 4;
    uint64_t *s_delta = reinterpret_cast<uint64_t *>(s_eI + 20);  // band deltas of this read, one bit per diagonal
    int *s_EF = reinterpret_cast<int *>(s_delta + n_blk);
    uint8_t *s_x = reinterpret_cast<uint8_t *>(s_EF + n_blk);
    uint8_t *s_ey = s_x + ((lds_tmpl + 16) & ~15u);
    const int lane = threadIdx.x;
    double2 *scratch = reinterpret_cast<double2 *>(scratch_all + (uint64_t)blockIdx.x * scratch_stride);

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
        const int L = (int)st.tmpl_len, n = (int)rm.read_len, T = L + n, r = (int)cm.radius;
        const HmmDev *h = hmm2 + (rm.strand ? 0 : 1);
        const uint64_t *delta = delta_all + rm.delta_off;
        __syncthreads();
        // ---- stage the template codes, the read's emission indices and the emission tables in LDS
        {
            const uint8_t *gx = bufs.tmpl[st.buf] + cm.tmpl_off;
            for (int p = lane; p < L; p += 64) s_x[p] = gx[p];
            const uint8_t *gy = ey_all + rm.ey_off;
            for (int p = lane; p <= n; p += 64) s_ey[p] = gy[p];
            if (lane < 16) s_eM[lane] = h->eM[lane];
            if (lane < 20) s_eI[lane] = h->eI[lane];
            for (int wdx = lane; wdx < (T >> 6) + 2; wdx += 64) s_delta[wdx] = delta[wdx];
        }
        __syncthreads();
        const double aMM = h->a[0], aMI = h->a[1], aMD = h->a[2], aIM = h->a[3], aII = h->a[4], aID = h->a[5],
                     aDM = h->a[6], aDI = h->a[7], aDD = h->a[8];

        // =========================== forward ===========================
        int c = 0, EF = 0;
        double toM_1 = 0, toM_2 = 0, toI_1 = 0, toD_1 = 0;  // combos of diagonals t-1 / t-2, lane frame
        double endM = 0, endI = 0, endD = 0;
        for (int t = 0; t <= T; t++) {
            if (t > 0) c += delta_bit(s_delta, t);
            const Lane x = lane_cell(lane, c, r, t, L, n);
            double fm = 0, fi = 0, fd = 0;
            if (t == 0) {
                fm = (x.active && x.i == 0) ? 1.0 : 0.0;
            } else {
                const double pM = rot_from_prev(toM_2), pD = rot_from_prev(toD_1), pI = toI_1;
                const int jj = x.j < 1 ? 1 : (x.j > n ? n : x.j);
                const int ii = x.i < 1 ? 1 : (x.i > L ? L : x.i);
                const int eyv = s_ey[jj];
                const int xc = s_x[ii - 1];
                const double eMv = s_eM[4 * xc + (eyv & 3)], eIv = s_eI[eyv];
                if (x.active && x.j >= 1) {
                    fi = eIv * pI;
                    if (x.i >= 1) fm = eMv * pM;
                }
                if (x.active && x.i >= 1) fd = pD;
            }
            if (t > 0 && (t & (JTK_SCALE_BLOCK - 1)) == 0) {
                double m = fm > fi ? fm : fi;
                m = fd > m ? fd : m;
                m = wave_max(m);
                if (m > 0.0) {
                    const int e = jtk_ilogb_pos(m);
                    const double s = pow2i(-e);
                    fm *= s;
                    fi *= s;
                    fd *= s;
                    toM_1 *= s;
                    EF += e;
                }
            }
            if ((t & (JTK_SCALE_BLOCK - 1)) == 0 && lane == 0) s_EF[t >> 6] = EF;
            const double toM = fma(fd, aDM, fma(fi, aIM, fm * aMM));
            const double toI = fma(fd, aDI, fma(fi, aII, fm * aMI));
            const double toD = fma(fd, aDD, fma(fi, aID, fm * aMD));
            scratch[(uint64_t)t * 64 + lane] = make_double2(toM, toD);
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
        // cell (L, n) sits on the lane that owns row L
        const int lane_end = L & 63;
        double tot = (endM + endI) + endD;
        tot = __shfl(tot, lane_end, 64);
        const double lk = tot > 0.0 ? jtk_log(tot) + (double)EF * JTK_LN2 : JTK_LOG_ZERO;
        if (lane == 0) lk_all[item] = lk;
        __syncthreads();  // s_EF visible; forward stores are read back by this same wave below

        // =========================== backward + table accumulation ===========================
        double *raw = raw_all + rm.raw_off;
        int *rawG = rawG_all + rm.row_off;
        double acc[JTK_ACC_N];
#pragma unroll
        for (int k = 0; k < JTK_ACC_N; k++) acc[k] = 0.0;
        double hM_1 = 0, hM_2 = 0, hI_1 = 0, bD_1 = 0;  // hatM(t+1), hatM(t+2), hatI(t+1), b_D(t+1)
        int EB = 0, Gprev = 0;
        // c currently == c[T]; centres of the del-3 source diagonals t-5 / t-4 are tracked separately
        int c5 = c, c4 = c;  // will be set below
        {
            int cc = c;
            // c[T-4], c[T-5] by walking back
            int tt = T;
            for (int s = 0; s < 4 && tt >= 1; s++, tt--) cc -= delta_bit(s_delta, tt);
            c4 = cc;
            if (tt >= 1) cc -= delta_bit(s_delta, tt);
            c5 = cc;
        }
        // ring preload: diagonals T+2 .. T-5 (those > T are zero)
        for (int tt = T + 2; tt >= T - 5; tt--) {
            double2 v = make_double2(0.0, 0.0);
            if (tt >= 0 && tt <= T) v = scratch[(uint64_t)tt * 64 + lane];
            ring[(tt & 7) * 64 + lane] = v;
        }
        // prefetch queue: pf[u] holds diagonal (t - 5 - 1) for the step that will consume it
        double2 pf[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int tt = T - 6 - u;
            pf[u] = tt >= 0 ? scratch[(uint64_t)tt * 64 + lane] : make_double2(0.0, 0.0);
        }
        __syncthreads();
        int delta_next = 0;  // c[t+1] - c[t]
        for (int tbase = T; tbase >= 0; tbase -= PF) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int t = tbase - u;
                if (t < 0) break;
                // centres: c == c[t+1] on entry (or c[T] when t == T)
                if (t < T) c -= delta_next;
                const Lane x = lane_cell(lane, c, r, t, L, n);
                // (0) a row that left the band at this step is final (exponent of the previous step)
                {
                    const int lo = c - r;
                    const int off = (lane - lo) & 63;
                    const bool flush = t < T && delta_next == 1 && off == 2 * r + 1 && x.i >= 0 && x.i <= L;
                    if (flush) {
                        double2 *dst = reinterpret_cast<double2 *>(raw + (uint64_t)x.i * JTK_ACC_N);
#pragma unroll
                        for (int k = 0; k < JTK_ACC_N / 2; k++) dst[k] = make_double2(acc[2 * k], acc[2 * k + 1]);
                        rawG[x.i] = Gprev;
#pragma unroll
                        for (int k = 0; k < JTK_ACC_N; k++) acc[k] = 0.0;
                    }
                }
                // (1) backward values of this diagonal
                double vm = 0, vi = 0, vd = 0;
                if (t == T) {
                    vm = vi = vd = x.active ? 1.0 : 0.0;
                } else {
                    const double xm = rot_from_next(hM_2), xd = rot_from_next(bD_1), xi = hI_1;
                    if (x.active) {
                        vm = fma(aMD, xd, fma(aMI, xi, aMM * xm));
                        vi = fma(aID, xd, fma(aII, xi, aIM * xm));
                        vd = fma(aDD, xd, fma(aDI, xi, aDM * xm));
                    }
                }
                if (t < T && (t & (JTK_SCALE_BLOCK - 1)) == JTK_SCALE_BLOCK - 1) {
                    double m = vm > vi ? vm : vi;
                    m = vd > m ? vd : m;
                    m = wave_max(m);
                    if (m > 0.0) {
                        const int e = jtk_ilogb_pos(m);
                        const double s = pow2i(-e);
                        vm *= s;
                        vi *= s;
                        vd *= s;
                        hM_1 *= s;
                        EB += e;
                    }
                }
                const int jj = x.j < 1 ? 1 : (x.j > n ? n : x.j);
                const int ii = x.i < 1 ? 1 : (x.i > L ? L : x.i);
                const int eyv = s_ey[jj];
                const int xc = s_x[ii - 1];
                const int y1 = eyv & 3;
                double hM = 0, hI = 0;
                if (x.active && x.j >= 1) {
                    hI = s_eI[eyv] * vi;
                    if (x.i >= 1) hM = s_eM[4 * xc + y1] * vm;
                }
                // (2) common exponent of this step
                const int EFt = s_EF[t >> 6];
                const int G = EFt + EB;
                if (t < T && G != Gprev) {
                    const double s = pow2i(Gprev - G);
#pragma unroll
                    for (int k = 0; k < JTK_ACC_N; k++) acc[k] *= s;
                }
                Gprev = G;
                // (3) the 16 row-crossing products of this cell; F values come from the LDS ring with the
                //     lane offset of the source row folded into the address.  A source diagonal in another
                //     64-diagonal block carries another exponent and is re-expressed by an exact power of two; the
                //     window t-5 .. t+2 lies inside one block for 57 of 64 steps, and then every factor is 1.
                const bool straddle = ((t - 5) >> 6) != ((t + 2) >> 6);
                auto products = [&](auto straddle_tag) {
                    constexpr bool ST = decltype(straddle_tag)::value;
                    auto FS = [&](int tt) -> double {  // 2^(EF[tt]-EF[t])
                        if (!ST) return 1.0;
                        if (tt < 0 || tt > T) return 1.0;
                        return fast_pow2(s_EF[tt >> 6] - EFt);
                    };
                    const bool jm = x.active && x.j >= 1;  // M terms consume read base y[j-1]
                    const bool i1 = x.active && x.i >= 1;
                    // sub (entry i-1): toM(i-1, j-1) on t-2, toD(i-1, j) on t-1
                    {
                        const double2 a = ring[((t - 2) & 7) * 64 + ((lane - 1) & 63)];
                        const double2 b = ring[((t - 1) & 7) * 64 + ((lane - 1) & 63)];
                        const double fmv = (i1 && jm && t >= 2) ? (ST ? a.x * FS(t - 2) : a.x) : 0.0;
                        const double fdv = (i1 && t >= 1) ? (ST ? b.y * FS(t - 1) : b.y) : 0.0;
#pragma unroll
                        for (int q = 0; q < 4; q++) acc[q] = fma(y1 == q ? fmv : 0.0, vm, acc[q]);
                        acc[4] = fma(fdv, vd, acc[4]);
                    }
                    // ins (entry i): toM(i, j-1) on t-1, toD(i, j) on t
                    {
                        const double2 a = ring[((t - 1) & 7) * 64 + lane];
                        const double2 b = ring[(t & 7) * 64 + lane];
                        const double fmv = (jm && t >= 1) ? (ST ? a.x * FS(t - 1) : a.x) : 0.0;
                        const double fdv = x.active ? b.y : 0.0;
#pragma unroll
                        for (int q = 0; q < 4; q++) acc[5 + q] = fma(y1 == q ? fmv : 0.0, vm, acc[5 + q]);
                        acc[9] = fma(fdv, vd, acc[9]);
                    }
                    // copy c (entry i-1): toM(i-1+c, j-1) on t+c-2, toD(i-1+c, j) on t+c-1
#pragma unroll
                    for (int cc = 1; cc <= 3; cc++) {
                        const double2 a = ring[((t + cc - 2) & 7) * 64 + ((lane - 1 + cc) & 63)];
                        const double2 b = ring[((t + cc - 1) & 7) * 64 + ((lane - 1 + cc) & 63)];
                        const double fmv = (i1 && jm && t + cc - 2 >= 0 && t + cc - 2 <= T) ? (ST ? a.x * FS(t + cc - 2) : a.x) : 0.0;
                        const double fdv = (i1 && t + cc - 1 <= T) ? (ST ? b.y * FS(t + cc - 1) : b.y) : 0.0;
                        double v = acc[10 + cc - 1];
                        v = fma(fmv, hM, v);
                        v = fma(fdv, vd, v);
                        acc[10 + cc - 1] = v;
                    }
                    // del d (entry i-d-1): toM(i-d-1, j-1) on t-d-2, toD(i-d-1, j) on t-d-1
#pragma unroll
                    for (int dd = 1; dd <= 3; dd++) {
                        const double2 a = ring[((t - dd - 2) & 7) * 64 + ((lane - dd - 1) & 63)];
                        const double2 b = ring[((t - dd - 1) & 7) * 64 + ((lane - dd - 1) & 63)];
                        bool okm = i1 && jm && x.i - dd - 1 >= 0 && t - dd - 2 >= 0;
                        bool okd = i1 && x.i - dd - 1 >= 0 && t - dd - 1 >= 0;
                        if (dd == 3) {  // the only source row the 3 spare lanes cannot disambiguate
                            okm = okm && (x.i - 4 >= c5 - r);
                            okd = okd && (x.i - 4 >= c4 - r);
                        }
                        const double fmv = okm ? (ST ? a.x * FS(t - dd - 2) : a.x) : 0.0;
                        const double fdv = okd ? (ST ? b.y * FS(t - dd - 1) : b.y) : 0.0;
                        double v = acc[13 + dd - 1];
                        v = fma(fmv, hM, v);
                        v = fma(fdv, vd, v);
                        acc[13 + dd - 1] = v;
                    }
                };
                if (straddle)
                    products(std::true_type{});
                else
                    products(std::false_type{});
                __syncthreads();
                // slide: diagonal t-6 replaces diagonal t+2 in the ring; refill the queue slot
                ring[((t - 6) & 7) * 64 + lane] = pf[u];
                {
                    const int tt = t - 6 - PF;
                    pf[u] = tt >= 0 ? scratch[(uint64_t)tt * 64 + lane] : make_double2(0.0, 0.0);
                }
                __syncthreads();
                hM_2 = hM_1;
                hM_1 = hM;
                hI_1 = hI;
                bD_1 = vd;
                // centres for the next step (t-1): c[t] -> c[t-1], c5 = c[t-6], c4 = c[t-5]
                delta_next = t >= 1 ? delta_bit(s_delta, t) : 0;
                c4 = c5;
                if (t - 5 >= 1) c5 -= delta_bit(s_delta, t - 5);
            }
        }
        // rows still in the band after t == 0
        {
            const Lane x = lane_cell(lane, c, r, 0, L, n);
            const int off = (lane - (c - r)) & 63;
            if (off <= 2 * r && x.i >= 0 && x.i <= L) {
                double2 *dst = reinterpret_cast<double2 *>(raw + (uint64_t)x.i * JTK_ACC_N);
#pragma unroll
                for (int k = 0; k < JTK_ACC_N / 2; k++) dst[k] = make_double2(acc[2 * k], acc[2 * k + 1]);
                rawG[x.i] = Gprev;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// finalize: one thread per (read, position p): the 14 table entries of p from the raw row sums,
// MINUS the read's lk (pseudo_mcmc.rs:64).  Row iota owns sub[iota-1], ins[iota], copy_c[iota-1],
// del_d[iota-d-1].
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double fin_log(double v, int G, double lk) {
    return (v > 0.0 ? jtk_log(v) + (double)G * JTK_LN2 : JTK_LOG_ZERO) - lk;
}

__global__ void finalize_kernel(uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                                const ChunkState *state, const HmmDev *hmm2, const double *raw_all,
                                const int *rawG_all, const double *lk_all, double *table_all, int only_active) {
    const uint32_t item = blockIdx.y;
    const ReadMeta rm = reads[item];
    const ChunkState st = state[rm.chunk];
    if (st.status != 0) return;
    if (only_active && !st.active) return;
    const int L = (int)st.tmpl_len;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > L) return;
    const HmmDev *h = hmm2 + (rm.strand ? 0 : 1);
    const double *raw = raw_all + rm.raw_off;
    const int *rawG = rawG_all + rm.row_off;
    const double lk = lk_all[item];
    double *out = table_all + rm.table_off + (uint64_t)p * JTK_NUM_ROW;
    const bool dead = !(lk > JTK_LOG_ZERO);
    double res[JTK_NUM_ROW];
#pragma unroll
    for (int k = 0; k < JTK_NUM_ROW; k++) res[k] = JTK_LOG_ZERO - (dead ? 0.0 : lk);
    if (!dead) {
        if (p + 1 <= L) {  // row p+1: sub[p], copy_c[p]
            const double *a = raw + (uint64_t)(p + 1) * JTK_ACC_N;
            const int G = rawG[p + 1];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                double v = h->eM[4 * b + 0] * a[0];
                v = fma(h->eM[4 * b + 1], a[1], v);
                v = fma(h->eM[4 * b + 2], a[2], v);
                v = fma(h->eM[4 * b + 3], a[3], v);
                v = v + a[4];
                res[b] = fin_log(v, G, lk);
            }
#pragma unroll
            for (int cc = 0; cc < 3; cc++) res[8 + cc] = fin_log(a[10 + cc], G, lk);
        }
        {  // row p: ins[p]
            const double *a = raw + (uint64_t)p * JTK_ACC_N;
            const int G = rawG[p];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                double v = h->eM[4 * b + 0] * a[5];
                v = fma(h->eM[4 * b + 1], a[6], v);
                v = fma(h->eM[4 * b + 2], a[7], v);
                v = fma(h->eM[4 * b + 3], a[8], v);
                v = v + a[9];
                res[4 + b] = fin_log(v, G, lk);
            }
        }
#pragma unroll
        for (int dd = 1; dd <= 3; dd++) {  // row p+d+1: del_d[p]
            if (p + dd + 1 <= L) {
                const double *a = raw + (uint64_t)(p + dd + 1) * JTK_ACC_N;
                res[11 + dd - 1] = fin_log(a[13 + dd - 1], rawG[p + dd + 1], lk);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < JTK_NUM_ROW; k++) out[k] = res[k];
}

}  // namespace

void launch_band_prep(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                      ChunkState *state, DevBufs bufs, uint64_t *delta, int only_active) {
    if (n_reads == 0) return;
    band_prep_kernel<<<(n_reads + 63) / 64, 64, 0, s>>>(n_reads, reads, chunks, state, bufs, delta, only_active);
}

size_t phmm_lds_bytes(uint32_t max_tmpl, uint32_t max_read) {
    const uint32_t n_blk = ((max_tmpl + max_read) >> 6) + 4;
    size_t b = 8 * 64 * 16 + 36 * 8 + (size_t)n_blk * 12;
    b += ((max_tmpl + 16) & ~15u) + max_read + 16;
    return (b + 15) & ~(size_t)15;
}

void launch_phmm(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                 const ChunkState *state, DevBufs bufs, const uint8_t *ey, const uint64_t *delta,
                 const HmmDev *hmm2, double *scratch, uint64_t scratch_stride, uint32_t n_waves,
                 uint32_t *work_counter, double *raw, int *rawG, double *lk, uint32_t max_tmpl,
                 uint32_t max_read, int only_active) {
    if (n_reads == 0) return;
    hipMemsetAsync(work_counter, 0, sizeof(uint32_t), s);
    const size_t lds = phmm_lds_bytes(max_tmpl, max_read);
    phmm_kernel<<<n_waves, 64, lds, s>>>(n_reads, reads, chunks, state, bufs, ey, delta, hmm2, scratch,
                                         scratch_stride, work_counter, raw, rawG, lk, max_tmpl, max_read,
                                         only_active);
}

void launch_finalize(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                     const ChunkState *state, const HmmDev *hmm2, const double *raw, const int *rawG,
                     const double *lk, double *table, uint32_t max_tmpl, int only_active) {
    if (n_reads == 0) return;
    dim3 grid((max_tmpl + 1 + 127) / 128, n_reads);
    finalize_kernel<<<grid, 128, 0, s>>>(n_reads, reads, chunks, state, hmm2, raw, rawG, lk, table, only_active);
}
