// phmm_wide.hip -- banded pair-HMM forward/backward + modification-table row sums for bands WIDER than one wavefront.
//
// phmm_kernel (phmm_kernels.hip) maps one anti-diagonal of the band onto the 64 lanes of a wave ("lane ring"), which
// takes band radii up to 30.  ReadType::band_width (definitions/src/lib.rs:173-175,201-210) asks for more: CLR / None
// reads (fraction 0.05: radius 50 at 2 kbp) and ONT chunks longer than 2,033 bp (a consensus that grew in an earlier
// round).  This kernel takes those reads: same specification (DESIGN.md "Pair-HMM specification", oracle/phmm.c), same
// outputs (raw row sums + exponent per template row and lk per read, which finalize_kernel turns into the table), any
// radius up to JTK_WIDE_MAX_RADIUS, written for correctness and generality rather than speed -- it follows the oracle's
// band-offset formulation cell by cell, one wave per read, the band swept in passes of 64 cells:
//  * forward: diagonals t-1 / t-2 of (toM, toI, toD) live in LDS rings; toM and toD of every diagonal go to a per-wave
//    scratch stripe in HBM ((T+1) x W doubles each) for the backward sweep;
//  * backward: three diagonals of (hatM, hatI, b_D) in LDS rings; each band cell adds its 16 row-crossing products to the
//    accumulators of ITS template row, kept in an LDS ring indexed by row (a row is touched by one cell per diagonal);
//    a row that leaves the band is written out with the exponent of the step before, exactly as phmm_kernel does.
// Every sum has the order phmm_kernel and the oracle use (j descending, M-term before D-term), every scale factor is an
// exact power of two, so the three agree bit for bit (tests/test_gpu_shapes.py::test_wide_band_*).
#include "device_common.h"

// Band cells per LDS ring row: a power of two >= 2 * radius + 1, chosen per launch from the largest radius of the batch (256 up
// to radius 127; 512 up to JTK_WIDE_MAX_RADIUS = 255 -- round 6: CLR / None reads on chunks up to 10 kbp, ONT up to 17 kbp,
// ReadType::band_width definitions/src/lib.rs:201-210).  A kernel argument: `WP` below is a local of each kernel.
static uint32_t wide_wp(uint32_t max_radius) { return max_radius <= 127u ? 256u : 512u; }

namespace {

__device__ __forceinline__ double pow2i_w(int e) { return jtk_scalbn(1.0, e); }
__device__ __forceinline__ double wave_max_w(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}

__global__ __launch_bounds__(64) void phmm_wide_kernel(uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                                                       const ChunkState *state, DevBufs bufs, const uint8_t *ey_all,
                                                       const uint64_t *delta_all, const HmmDev *hmm2, double *scratch_all,
                                                       uint64_t scratch_stride, uint32_t *work_counter, uint32_t ticket_base, double *raw_all,
                                                       int *rawG_all, double *lk_all, uint32_t lds_tmpl, uint32_t lds_read,
                                                       int only_active, uint32_t wp) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int WP = (int)__builtin_amdgcn_readfirstlane(wp);
    // LDS carve: rings (9 rows of WP doubles, shared by the two sweeps) | acc ring [WP][16] | EF / EB per 64-block |
    // centres c[t] (u16) | template codes | read codes
    double *ring = reinterpret_cast<double *>(smem);          // forward: toM x3, toI x2, toD x2; backward: hM, hI, bD x3
    double *acc = ring + 9 * WP;                              // [row & (WP-1)][16]
    const uint32_t n_blk = ((lds_tmpl + lds_read) >> 6) + 4;
    int *s_EF = reinterpret_cast<int *>(acc + WP * JTK_ACC_N);
    int *s_EB = s_EF + n_blk;
    uint16_t *s_c = reinterpret_cast<uint16_t *>(s_EB + n_blk);
    uint8_t *s_x = reinterpret_cast<uint8_t *>(s_c + ((lds_tmpl + lds_read + 8) & ~7u));
    uint8_t *s_y = s_x + ((lds_tmpl + 16) & ~15u);
    const int lane = threadIdx.x;

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
        if (cm.radius <= JTK_MAX_RADIUS) continue;  // phmm_kernel's read
        if (cm.take_num && item - cm.read_first >= cm.take_num) continue;
        const int L = (int)st.tmpl_len, n = (int)rm.read_len, T = L + n, r = (int)cm.radius, W = 2 * r + 1;
        const int NP = (W + 63) >> 6;
        const HmmDev *h = hmm2 + (rm.strand ? 0 : 1);
        const double aMM = h->a[0], aMI = h->a[1], aMD = h->a[2], aIM = h->a[3], aII = h->a[4], aID = h->a[5], aDM = h->a[6],
                     aDI = h->a[7], aDD = h->a[8];
        double *fM = scratch_all + (uint64_t)blockIdx.x * scratch_stride;  // toM[t][w]
        double *fD = fM + (uint64_t)(T + 1) * W;                           // toD[t][w]
        __syncthreads();
        // ---- stage codes and the centres c[t] (prefix sums of band_prep's delta bits)
        {
            const uint8_t *gx = bufs.tmpl[st.buf] + cm.tmpl_off;
            for (int p = lane; p < L; p += 64) s_x[p] = gx[p] & 3;
            const uint8_t *gy = ey_all + rm.ey_off;
            for (int p = lane; p < n; p += 64) s_y[p] = gy[p + 1] & 3;
            const uint64_t *delta = delta_all + rm.delta_off;
            if (lane == 0) {
                uint32_t c = 0;
                for (int t = 0; t <= T; t++) {
                    if (t >= 1) c += (uint32_t)((delta[t >> 6] >> (t & 63)) & 1ull);
                    s_c[t] = (uint16_t)c;
                }
            }
        }
        __syncthreads();
        auto eMv = [&](int i, int j) -> double { return h->eM[4 * s_x[i - 1] + s_y[j - 1]]; };          // i, j >= 1
        auto eIv = [&](int j) -> double { return h->eI[4 * (j >= 2 ? (int)s_y[j - 2] : 4) + s_y[j - 1]]; };  // j >= 1
        // ring rows: a diagonal tt of array A lives in row base_A + tt % depth, indexed by its own band offset
        double *rM = ring, *rI = ring + 3 * WP, *rD = ring + 5 * WP;  // forward: toM (depth 3), toI (2), toD (2)
        auto getw = [&](const double *row, int tt, int i) -> double {  // value of row `i` on diagonal tt (0 outside the band)
            if (tt < 0 || tt > T) return 0.0;
            const int w = i - ((int)s_c[tt] - r);
            return (w < 0 || w >= W) ? 0.0 : row[w];
        };

        // =========================== forward ===========================
        int E = 0;  // cumulative exponent of the current diagonal
        double endM = 0, endI = 0, endD = 0;
        for (int t = 0; t <= T; t++) {
            const int Eprev1 = E, Eprev2 = t >= 2 ? s_EF[(t - 2) >> 6] : 0;
            (void)Eprev1;
            // diagonal t-2 is re-expressed in the scale of diagonal t-1 (they differ only across a block boundary)
            const int E1 = t >= 1 ? s_EF[(t - 1) >> 6] : 0;
            const double s2 = (t >= 2 && Eprev2 != E1) ? pow2i_w(Eprev2 - E1) : 1.0;
            const int lo = (int)s_c[t] - r;
            double fm[4], fi[4], fd[4];
            double m = 0.0;
            for (int ps = 0; ps < NP; ps++) {
                const int w = ps * 64 + lane, i = lo + w, j = t - i;
                double a = 0.0, b = 0.0, c = 0.0;
                if (w < W && i >= 0 && i <= L && j >= 0 && j <= n) {
                    if (t == 0) {
                        a = 1.0;
                    } else {
                        if (i >= 1 && j >= 1) a = eMv(i, j) * (getw(rM + ((t - 2 + 3) % 3) * WP, t - 2, i - 1) * s2);
                        if (j >= 1) b = eIv(j) * getw(rI + ((t - 1) & 1) * WP, t - 1, i);
                        if (i >= 1) c = getw(rD + ((t - 1) & 1) * WP, t - 1, i - 1);
                    }
                }
                fm[ps] = a;
                fi[ps] = b;
                fd[ps] = c;
                m = a > m ? a : m;
                m = b > m ? b : m;
                m = c > m ? c : m;
            }
            E = E1;
            if (t > 0 && (t & (JTK_SCALE_BLOCK - 1)) == 0) {
                m = wave_max_w(m);
                if (m > 0.0) {
                    const int e = jtk_ilogb_pos(m);
                    const double sc = pow2i_w(-e);
                    for (int ps = 0; ps < NP; ps++) {
                        fm[ps] *= sc;
                        fi[ps] *= sc;
                        fd[ps] *= sc;
                    }
                    E += e;
                }
            }
            if (lane == 0 && (t & 63) == 0) s_EF[t >> 6] = E;
            __syncthreads();  // everyone has read diagonals t-1 / t-2 before their ring rows are reused
            double *oM = rM + (t % 3) * WP, *oI = rI + (t & 1) * WP, *oD = rD + (t & 1) * WP;
            for (int ps = 0; ps < NP; ps++) {
                const int w = ps * 64 + lane;
                if (w < W) {
                    const double toM = fma(fd[ps], aDM, fma(fi[ps], aIM, fm[ps] * aMM));
                    const double toI = fma(fd[ps], aDI, fma(fi[ps], aII, fm[ps] * aMI));
                    const double toD = fma(fd[ps], aDD, fma(fi[ps], aID, fm[ps] * aMD));
                    oM[w] = toM;
                    oI[w] = toI;
                    oD[w] = toD;
                    fM[(uint64_t)t * W + w] = toM;
                    fD[(uint64_t)t * W + w] = toD;
                    if (t == T && w == r) {  // c[T] == L: offset r is cell (L, n)
                        endM = fm[ps];
                        endI = fi[ps];
                        endD = fd[ps];
                    }
                }
            }
            __syncthreads();
        }
        double tot = (endM + endI) + endD;  // held by the lane that owns offset r
        tot = __shfl(tot, r & 63, 64);
        const int ET = s_EF[T >> 6];
        const double lk = tot > 0.0 ? jtk_log(tot) + (double)ET * JTK_LN2 : JTK_LOG_ZERO;
        if (lane == 0) lk_all[item] = lk;
        __syncthreads();  // forward stores are read back by this same wave below

        // =========================== backward + table accumulation ===========================
        double *raw = raw_all + rm.raw_off;
        int *rawG = rawG_all + rm.row_off;
        for (int e = lane; e < WP * JTK_ACC_N; e += 64) acc[e] = 0.0;
        double *bHM = ring, *bHI = ring + 3 * WP, *bBD = ring + 6 * WP;  // hatM, hatI, b_D: depth 3 each
        for (int e = lane; e < 9 * WP; e += 64) ring[e] = 0.0;
        __syncthreads();
        auto fget = [&](const double *arr, int tt, int i) -> double {  // forward table with its own band offsets
            if (tt < 0 || tt > T) return 0.0;
            const int w = i - ((int)s_c[tt] - r);
            return (w < 0 || w >= W) ? 0.0 : arr[(uint64_t)tt * W + w];
        };
        int EB = 0, Gprev = 0, live_hi = L;
        for (int t = T; t >= 0; t--) {
            const int lo = (int)s_c[t] - r, hi = lo + W - 1;
            // (0) rows that left the band are final, in the exponent of the previous step
            for (int i = live_hi - lane; i > hi; i -= 64) {
                double *dst = raw + (uint64_t)i * JTK_ACC_N;
                double *a = acc + (i & (WP - 1)) * JTK_ACC_N;
                for (int k = 0; k < JTK_ACC_N; k++) {
                    dst[k] = a[k];
                    a[k] = 0.0;
                }
                rawG[i] = Gprev;
            }
            if (live_hi > hi) live_hi = hi;
            __syncthreads();
            // (1) backward values of diagonal t, produced in the scale of diagonal t+1
            const int Ecur0 = t < T ? s_EB[(t + 1) >> 6] : 0;
            const int E2 = t + 2 <= T ? s_EB[(t + 2) >> 6] : 0;
            const double s2 = (t + 2 <= T && E2 != Ecur0) ? pow2i_w(E2 - Ecur0) : 1.0;
            double vm[4], vi[4], vd[4];
            double m = 0.0;
            for (int ps = 0; ps < NP; ps++) {
                const int w = ps * 64 + lane, i = lo + w, j = t - i;
                double a = 0.0, b = 0.0, c = 0.0;
                if (w < W && i >= 0 && i <= L && j >= 0 && j <= n) {
                    if (t == T) {
                        a = b = c = 1.0;
                    } else {
                        const double xm = getw(bHM + ((t + 2) % 3) * WP, t + 2, i + 1) * s2;
                        const double xi = getw(bHI + ((t + 1) % 3) * WP, t + 1, i);
                        const double xd = getw(bBD + ((t + 1) % 3) * WP, t + 1, i + 1);
                        a = fma(aMD, xd, fma(aMI, xi, aMM * xm));
                        b = fma(aID, xd, fma(aII, xi, aIM * xm));
                        c = fma(aDD, xd, fma(aDI, xi, aDM * xm));
                    }
                }
                vm[ps] = a;
                vi[ps] = b;
                vd[ps] = c;
                m = a > m ? a : m;
                m = b > m ? b : m;
                m = c > m ? c : m;
            }
            EB = Ecur0;
            if (t < T && (t & (JTK_SCALE_BLOCK - 1)) == JTK_SCALE_BLOCK - 1) {
                m = wave_max_w(m);
                if (m > 0.0) {
                    const int e = jtk_ilogb_pos(m);
                    const double sc = pow2i_w(-e);
                    for (int ps = 0; ps < NP; ps++) {
                        vm[ps] *= sc;
                        vi[ps] *= sc;
                        vd[ps] *= sc;
                    }
                    EB += e;
                }
            }
            if (lane == 0) s_EB[t >> 6] = EB;  // (the block's value: set by its highest diagonal, kept by the others)
            __syncthreads();
            double *oHM = bHM + (t % 3) * WP, *oHI = bHI + (t % 3) * WP, *oBD = bBD + (t % 3) * WP;
            double hm[4];
            for (int ps = 0; ps < NP; ps++) {
                const int w = ps * 64 + lane, i = lo + w, j = t - i;
                double a = 0.0, b = 0.0;
                if (w < W) {
                    if (i >= 1 && i <= L && j >= 1 && j <= n) a = eMv(i, j) * vm[ps];
                    if (i >= 0 && i <= L && j >= 1 && j <= n) b = eIv(j) * vi[ps];
                    oHM[w] = a;
                    oHI[w] = b;
                    oBD[w] = vd[ps];
                }
                hm[ps] = a;
            }
            // (2) common exponent of this step; live accumulators are re-expressed when it changes
            const int EFt = s_EF[t >> 6];
            const int G = EFt + EB;
            if (t < T && G != Gprev) {
                const double sc = pow2i_w(Gprev - G);
                const int from = lo < 0 ? 0 : lo;
                for (int i = from + lane; i <= live_hi; i += 64) {
                    double *a = acc + (i & (WP - 1)) * JTK_ACC_N;
                    for (int k = 0; k < JTK_ACC_N; k++) a[k] *= sc;
                }
            }
            Gprev = G;
            __syncthreads();
            // (3) the 16 row-crossing products of every band cell (row i, column j)
            double fsc[8];  // 2^(E_F[tt] - E_F[t]) for tt = t-5 .. t+2
            for (int q = 0; q < 8; q++) {
                const int tt = t - 5 + q;
                const int de = (tt >= 0 && tt <= T) ? s_EF[tt >> 6] - EFt : 0;
                fsc[q] = de == 0 ? 1.0 : pow2i_w(de);
            }
            for (int ps = 0; ps < NP; ps++) {
                const int w = ps * 64 + lane, i = lo + w, j = t - i;
                if (!(w < W && i >= 0 && i <= L && j >= 0 && j <= n)) continue;
                double *a = acc + (i & (WP - 1)) * JTK_ACC_N;
                const double vM = vm[ps], vD = vd[ps], vH = hm[ps];
#define FSC(tt) fsc[(tt) - t + 5]
                if (i >= 1) {  // sub (entry i-1)
                    if (j >= 1) {
                        const int cy = s_y[j - 1];
                        a[cy] = fma(fget(fM, t - 2, i - 1) * FSC(t - 2), vM, a[cy]);
                    }
                    a[4] = fma(fget(fD, t - 1, i - 1) * FSC(t - 1), vD, a[4]);
                }
                if (j >= 1) {  // ins (entry i)
                    const int cy = s_y[j - 1];
                    a[5 + cy] = fma(fget(fM, t - 1, i) * FSC(t - 1), vM, a[5 + cy]);
                }
                a[9] = fma(fget(fD, t, i), vD, a[9]);
                if (i >= 1) {
                    for (int c = 1; c <= 3; c++) {  // copy c (entry i-1)
                        double v = a[10 + c - 1];
                        if (j >= 1) v = fma(fget(fM, t + c - 2, i - 1 + c) * FSC(t + c - 2), vH, v);
                        v = fma(fget(fD, t + c - 1, i - 1 + c) * FSC(t + c - 1), vD, v);
                        a[10 + c - 1] = v;
                    }
                    for (int d = 1; d <= 3; d++) {  // del d (entry i-d-1)
                        if (i - d - 1 < 0) continue;
                        double v = a[13 + d - 1];
                        if (j >= 1) v = fma(fget(fM, t - d - 2, i - d - 1) * FSC(t - d - 2), vH, v);
                        v = fma(fget(fD, t - d - 1, i - d - 1) * FSC(t - d - 1), vD, v);
                        a[13 + d - 1] = v;
                    }
                }
#undef FSC
            }
            __syncthreads();
        }
        // rows still in the band after t == 0
        for (int i = live_hi - lane; i >= 0; i -= 64) {
            double *dst = raw + (uint64_t)i * JTK_ACC_N;
            const double *a = acc + (i & (WP - 1)) * JTK_ACC_N;
            for (int k = 0; k < JTK_ACC_N; k++) dst[k] = a[k];
            rawG[i] = Gprev;
        }
    }
}


// ------------------------------------------------------------------------------------------------------
// Expected transition / emission counts of one read (the E-step of the stage's model refit, model_tune.rs:144-151;
// kiley's fit is not under /root/reference: own specification, oracle/model_fit.c).  Same sweep structure as
// phmm_wide_kernel; the forward sweep keeps F_M, F_I, F_D of every diagonal in the scratch stripe, the backward sweep
// adds every cell's posterior-weighted transitions and emissions to the partial sums of its lane (band offset mod 64),
// which lane 0 finally adds up in lane order.  out: 45 counts per read (9 transitions, mat_emit[16], ins_emit[20]) + lk.
// ------------------------------------------------------------------------------------------------------
#define NCNT 45
__global__ __launch_bounds__(64) void phmm_counts_kernel(uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                                                         const ChunkState *state, DevBufs bufs, const uint8_t *ey_all,
                                                         const uint64_t *delta_all, const HmmDev *hmm2, double *scratch_all,
                                                         uint64_t scratch_stride, uint32_t *work_counter, uint32_t ticket_base, double *counts_all,
                                                         double *lk_all, uint32_t lds_tmpl, uint32_t lds_read, uint32_t wp) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int WP = (int)__builtin_amdgcn_readfirstlane(wp);
    double *ring = reinterpret_cast<double *>(smem);
    double *part = ring + 9 * WP;  // [64][NCNT]
    const uint32_t n_blk = ((lds_tmpl + lds_read) >> 6) + 4;
    int *s_EF = reinterpret_cast<int *>(part + 64 * NCNT + 1);
    int *s_EB = s_EF + n_blk;
    uint16_t *s_c = reinterpret_cast<uint16_t *>(s_EB + n_blk);
    uint8_t *s_x = reinterpret_cast<uint8_t *>(s_c + ((lds_tmpl + lds_read + 8) & ~7u));
    uint8_t *s_y = s_x + ((lds_tmpl + 16) & ~15u);
    const int lane = threadIdx.x;
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(work_counter, 1u) - ticket_base;  // tickets: the counter is never reset
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_reads) break;
        const ReadMeta rm = reads[item];
        const ChunkMeta cm = chunks[rm.chunk];
        const ChunkState st = state[rm.chunk];
        double *out = counts_all + (uint64_t)item * NCNT;
        if (st.status != 0) {
            if (lane < NCNT) out[lane] = 0.0;
            if (lane == 0) lk_all[item] = JTK_LOG_ZERO;
            continue;
        }
        const int L = (int)st.tmpl_len, n = (int)rm.read_len, T = L + n, r = (int)cm.radius, W = 2 * r + 1;
        const int NP = (W + 63) >> 6;
        const HmmDev *h = hmm2 + (rm.strand ? 0 : 1);
        const double aMM = h->a[0], aMI = h->a[1], aMD = h->a[2], aIM = h->a[3], aII = h->a[4], aID = h->a[5], aDM = h->a[6],
                     aDI = h->a[7], aDD = h->a[8];
        double *gM = scratch_all + (uint64_t)blockIdx.x * scratch_stride;  // F_M[t][w]
        double *gI = gM + (uint64_t)(T + 1) * W, *gD = gI + (uint64_t)(T + 1) * W;
        __syncthreads();
        {
            const uint8_t *gx = bufs.tmpl[st.buf] + cm.tmpl_off;
            for (int p = lane; p < L; p += 64) s_x[p] = gx[p] & 3;
            const uint8_t *gy = ey_all + rm.ey_off;
            for (int p = lane; p < n; p += 64) s_y[p] = gy[p + 1] & 3;
            const uint64_t *delta = delta_all + rm.delta_off;
            if (lane == 0) {
                uint32_t c = 0;
                for (int t = 0; t <= T; t++) {
                    if (t >= 1) c += (uint32_t)((delta[t >> 6] >> (t & 63)) & 1ull);
                    s_c[t] = (uint16_t)c;
                }
            }
            for (int e = lane; e < 64 * NCNT; e += 64) part[e] = 0.0;
        }
        __syncthreads();
        auto eMv = [&](int i, int j) -> double { return h->eM[4 * s_x[i - 1] + s_y[j - 1]]; };
        auto eIv = [&](int j) -> double { return h->eI[4 * (j >= 2 ? (int)s_y[j - 2] : 4) + s_y[j - 1]]; };
        double *rM = ring, *rI = ring + 3 * WP, *rD = ring + 6 * WP;  // toM / toI / toD, then hatM / hatI / b_D: depth 3 each
        auto getw = [&](const double *row, int tt, int i) -> double {
            if (tt < 0 || tt > T) return 0.0;
            const int w = i - ((int)s_c[tt] - r);
            return (w < 0 || w >= W) ? 0.0 : row[w];
        };
        // =========================== forward ===========================
        double endM = 0, endI = 0, endD = 0;
        for (int t = 0; t <= T; t++) {
            const int E2 = t >= 2 ? s_EF[(t - 2) >> 6] : 0, E1 = t >= 1 ? s_EF[(t - 1) >> 6] : 0;
            const double s2 = (t >= 2 && E2 != E1) ? pow2i_w(E2 - E1) : 1.0;
            const int lo = (int)s_c[t] - r;
            double fm[4], fi[4], fd[4];
            double m = 0.0;
            for (int ps = 0; ps < NP; ps++) {
                const int w = ps * 64 + lane, i = lo + w, j = t - i;
                double a = 0.0, b = 0.0, c = 0.0;
                if (w < W && i >= 0 && i <= L && j >= 0 && j <= n) {
                    if (t == 0) {
                        a = 1.0;
                    } else {
                        if (i >= 1 && j >= 1) a = eMv(i, j) * (getw(rM + ((t - 2 + 3) % 3) * WP, t - 2, i - 1) * s2);
                        if (j >= 1) b = eIv(j) * getw(rI + ((t - 1) % 3) * WP, t - 1, i);
                        if (i >= 1) c = getw(rD + ((t - 1) % 3) * WP, t - 1, i - 1);
                    }
                }
                fm[ps] = a;
                fi[ps] = b;
                fd[ps] = c;
                m = a > m ? a : m;
                m = b > m ? b : m;
                m = c > m ? c : m;
            }
            int E = E1;
            if (t > 0 && (t & (JTK_SCALE_BLOCK - 1)) == 0) {
                m = wave_max_w(m);
                if (m > 0.0) {
                    const int e = jtk_ilogb_pos(m);
                    const double sc = pow2i_w(-e);
                    for (int ps = 0; ps < NP; ps++) {
                        fm[ps] *= sc;
                        fi[ps] *= sc;
                        fd[ps] *= sc;
                    }
                    E += e;
                }
            }
            if (lane == 0 && (t & 63) == 0) s_EF[t >> 6] = E;
            __syncthreads();
            double *oM = rM + (t % 3) * WP, *oI = rI + (t % 3) * WP, *oD = rD + (t % 3) * WP;
            for (int ps = 0; ps < NP; ps++) {
                const int w = ps * 64 + lane;
                if (w < W) {
                    oM[w] = fma(fd[ps], aDM, fma(fi[ps], aIM, fm[ps] * aMM));
                    oI[w] = fma(fd[ps], aDI, fma(fi[ps], aII, fm[ps] * aMI));
                    oD[w] = fma(fd[ps], aDD, fma(fi[ps], aID, fm[ps] * aMD));
                    gM[(uint64_t)t * W + w] = fm[ps];
                    gI[(uint64_t)t * W + w] = fi[ps];
                    gD[(uint64_t)t * W + w] = fd[ps];
                    if (t == T && w == r) {
                        endM = fm[ps];
                        endI = fi[ps];
                        endD = fd[ps];
                    }
                }
            }
            __syncthreads();
        }
        double tot = (endM + endI) + endD;
        tot = __shfl(tot, r & 63, 64);
        const int ET = s_EF[T >> 6];
        const double lk = tot > 0.0 ? jtk_log(tot) + (double)ET * JTK_LN2 : JTK_LOG_ZERO;
        if (lane == 0) lk_all[item] = lk;
        if (!(tot > 0.0)) {
            if (lane < NCNT) out[lane] = 0.0;
            continue;
        }
        // =========================== backward + counts ===========================
        for (int e = lane; e < 9 * WP; e += 64) ring[e] = 0.0;
        __syncthreads();
        const double inv = 1.0 / tot;
        double *mine = part + lane * NCNT;
        for (int t = T; t >= 0; t--) {
            const int lo = (int)s_c[t] - r;
            const int Ecur0 = t < T ? s_EB[(t + 1) >> 6] : 0;
            const int E2 = t + 2 <= T ? s_EB[(t + 2) >> 6] : 0;
            const double s2 = (t + 2 <= T && E2 != Ecur0) ? pow2i_w(E2 - Ecur0) : 1.0;
            const int EFt = s_EF[t >> 6];
            const double wt = pow2i_w(EFt + Ecur0 - ET) * inv;
            double vm[4], vi[4], vd[4];
            double m = 0.0;
            for (int ps = 0; ps < NP; ps++) {
                const int w = ps * 64 + lane, i = lo + w, j = t - i;
                double a = 0.0, b = 0.0, c = 0.0;
                if (w < W && i >= 0 && i <= L && j >= 0 && j <= n) {
                    if (t == T) {
                        a = b = c = 1.0;
                    } else {
                        const double xm = getw(rM + ((t + 2) % 3) * WP, t + 2, i + 1) * s2;
                        const double xi = getw(rI + ((t + 1) % 3) * WP, t + 1, i);
                        const double xd = getw(rD + ((t + 1) % 3) * WP, t + 1, i + 1);
                        a = fma(aMD, xd, fma(aMI, xi, aMM * xm));
                        b = fma(aID, xd, fma(aII, xi, aIM * xm));
                        c = fma(aDD, xd, fma(aDI, xi, aDM * xm));
                        const double fm = gM[(uint64_t)t * W + w], fi = gI[(uint64_t)t * W + w], fd = gD[(uint64_t)t * W + w];
                        mine[0] += ((fm * aMM) * xm) * wt;
                        mine[1] += ((fm * aMI) * xi) * wt;
                        mine[2] += ((fm * aMD) * xd) * wt;
                        mine[3] += ((fi * aIM) * xm) * wt;
                        mine[4] += ((fi * aII) * xi) * wt;
                        mine[5] += ((fi * aID) * xd) * wt;
                        mine[6] += ((fd * aDM) * xm) * wt;
                        mine[7] += ((fd * aDI) * xi) * wt;
                        mine[8] += ((fd * aDD) * xd) * wt;
                    }
                }
                vm[ps] = a;
                vi[ps] = b;
                vd[ps] = c;
                m = a > m ? a : m;
                m = b > m ? b : m;
                m = c > m ? c : m;
            }
            int EB = Ecur0;
            if (t < T && (t & (JTK_SCALE_BLOCK - 1)) == JTK_SCALE_BLOCK - 1) {
                m = wave_max_w(m);
                if (m > 0.0) {
                    const int e = jtk_ilogb_pos(m);
                    const double sc = pow2i_w(-e);
                    for (int ps = 0; ps < NP; ps++) {
                        vm[ps] *= sc;
                        vi[ps] *= sc;
                        vd[ps] *= sc;
                    }
                    EB += e;
                }
            }
            if (lane == 0) s_EB[t >> 6] = EB;
            const double we = pow2i_w(EFt + EB - ET) * inv;
            __syncthreads();
            double *oM = rM + (t % 3) * WP, *oI = rI + (t % 3) * WP, *oD = rD + (t % 3) * WP;
            for (int ps = 0; ps < NP; ps++) {
                const int w = ps * 64 + lane, i = lo + w, j = t - i;
                if (w >= W) continue;
                double a = 0.0, b = 0.0;
                if (i >= 1 && i <= L && j >= 1 && j <= n) {
                    a = eMv(i, j) * vm[ps];
                    mine[9 + 4 * s_x[i - 1] + s_y[j - 1]] += (gM[(uint64_t)t * W + w] * vm[ps]) * we;
                }
                if (i >= 0 && i <= L && j >= 1 && j <= n) {
                    const int ctx = j >= 2 ? (int)s_y[j - 2] : 4;
                    b = eIv(j) * vi[ps];
                    mine[25 + 4 * ctx + s_y[j - 1]] += (gI[(uint64_t)t * W + w] * vi[ps]) * we;
                }
                oM[w] = a;
                oI[w] = b;
                oD[w] = vd[ps];
            }
            __syncthreads();
        }
        if (lane < NCNT) {  // the 64 partial sums, in lane order
            double sum = 0.0;
            for (int l = 0; l < 64; l++) sum += part[l * NCNT + lane];
            out[lane] = sum;
        }
    }
}
#undef NCNT

}  // namespace

size_t phmm_wide_lds_bytes(uint32_t max_tmpl, uint32_t max_read, uint32_t max_radius) {
    const uint32_t n_blk = ((max_tmpl + max_read) >> 6) + 4;
    const size_t WP = wide_wp(max_radius);
    size_t b = (size_t)9 * WP * 8 + (size_t)WP * JTK_ACC_N * 8 + (size_t)n_blk * 8;
    b += (size_t)((max_tmpl + max_read + 8) & ~7u) * 2 + ((max_tmpl + 16) & ~15u) + max_read + 16;
    return (b + 15) & ~(size_t)15;
}

// doubles of forward scratch one wave needs for a read of template length L, read length n, radius r
uint64_t phmm_wide_scratch_doubles(uint32_t max_tmpl, uint32_t max_read, uint32_t max_radius) {
    return (uint64_t)2 * (max_tmpl + max_read + 2) * (2 * max_radius + 1);
}

void launch_phmm_wide(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                      const ChunkState *state, DevBufs bufs, const uint8_t *ey, const uint64_t *delta, const HmmDev *hmm2,
                      double *scratch, uint64_t scratch_stride, uint32_t n_waves, uint32_t *work_counter, uint32_t *ticket_base,
                      double *raw, int *rawG, double *lk, uint32_t max_tmpl, uint32_t max_read, int only_active,
                      uint32_t max_radius) {
    if (n_reads == 0 || n_waves == 0) return;
    const uint32_t base = *ticket_base;
    const size_t lds = phmm_wide_lds_bytes(max_tmpl, max_read, max_radius);
    phmm_wide_kernel<<<n_waves, 64, lds, s>>>(n_reads, reads, chunks, state, bufs, ey, delta, hmm2, scratch, scratch_stride,
                                              work_counter, base, raw, rawG, lk, max_tmpl, max_read, only_active, wide_wp(max_radius));
    // the host mirror of the never-reset ticket counter moves only when the launch was accepted: a rejected launch (LDS, grid
    // or an earlier sticky error) takes no tickets, and a mirror that ran ahead would make every later launch exit at once
    if (hipPeekAtLastError() == hipSuccess) *ticket_base = base + n_reads + n_waves;
}

size_t phmm_counts_lds_bytes(uint32_t max_tmpl, uint32_t max_read, uint32_t max_radius) {
    const uint32_t n_blk = ((max_tmpl + max_read) >> 6) + 4;
    const size_t WP = wide_wp(max_radius);
    size_t b = (size_t)9 * WP * 8 + (size_t)(64 * 45 + 1) * 8 + (size_t)n_blk * 8;
    b += (size_t)((max_tmpl + max_read + 8) & ~7u) * 2 + ((max_tmpl + 16) & ~15u) + max_read + 16;
    return (b + 15) & ~(size_t)15;
}
uint64_t phmm_counts_scratch_doubles(uint32_t max_tmpl, uint32_t max_read, uint32_t max_radius) {
    return (uint64_t)3 * (max_tmpl + max_read + 2) * (2 * max_radius + 1);
}
void launch_phmm_counts(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                        const ChunkState *state, DevBufs bufs, const uint8_t *ey, const uint64_t *delta, const HmmDev *hmm2,
                        double *scratch, uint64_t scratch_stride, uint32_t n_waves, uint32_t *work_counter, uint32_t *ticket_base,
                        double *counts, double *lk, uint32_t max_tmpl, uint32_t max_read, uint32_t max_radius) {
    if (n_reads == 0 || n_waves == 0) return;
    const uint32_t base = *ticket_base;
    const size_t lds = phmm_counts_lds_bytes(max_tmpl, max_read, max_radius);
    phmm_counts_kernel<<<n_waves, 64, lds, s>>>(n_reads, reads, chunks, state, bufs, ey, delta, hmm2, scratch, scratch_stride,
                                                work_counter, base, counts, lk, max_tmpl, max_read, wide_wp(max_radius));
    // the host mirror of the never-reset ticket counter moves only when the launch was accepted: a rejected launch (LDS, grid
    // or an earlier sticky error) takes no tickets, and a mirror that ran ahead would make every later launch exit at once
    if (hipPeekAtLastError() == hipSuccess) *ticket_base = base + n_reads + n_waves;
}
