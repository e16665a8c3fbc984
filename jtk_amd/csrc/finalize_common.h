// finalize_common.h -- the arithmetic that turns a read's raw row sums into the 14 modification-table entries of a position
// (own specification of kiley's table; pseudo_mcmc.rs:64 subtracts the read's lk).  Shared by
// finalize_kernel / sum_final_kernel (phmm_kernels.hip) and, since round 6, by the variant filter (filter_kernels.hip), which
// takes its column statistics straight from the row sums: every user evaluates THESE expressions, so an entry has the same bits
// wherever it is computed.
#pragma once
#include "device_common.h"

namespace {

__device__ __forceinline__ double fin_log(double v, int G, double lk) {
    return (v > 0.0 ? jtk_log(v) + (double)G * JTK_LN2 : JTK_LOG_ZERO) - lk;
}

#define FIN_TILE 128                   // positions per tile
#define FIN_ROWS (FIN_TILE + 4)        // raw rows a tile reads: p .. p+4 for its last position
#define FIN_PITCH (JTK_ACC_N + 1)      // doubles per staged row: 17 keeps the 128-byte rows off each other's LDS banks
// rows p0 .. p0 + n_rows - 1 of a read's row sums into LDS, with coalesced 16-byte loads
__device__ __forceinline__ void fin_stage(double *s_raw, int *s_G, const double *raw, const int *rawG, int p0, int n_rows,
                                          int tid, bool dead) {
    if (dead) return;
    const double2 *src = reinterpret_cast<const double2 *>(raw + (uint64_t)p0 * JTK_ACC_N);
    for (int e = tid; e < n_rows * (JTK_ACC_N / 2); e += FIN_TILE) {
        const double2 v = src[e];
        const int row = e / (JTK_ACC_N / 2), k = e % (JTK_ACC_N / 2);
        s_raw[row * FIN_PITCH + 2 * k] = v.x;
        s_raw[row * FIN_PITCH + 2 * k + 1] = v.y;
    }
    for (int e = tid; e < n_rows; e += FIN_TILE) s_G[e] = rawG[p0 + e];
}
// the 14 table entries of position p (thread tid of the tile), minus the read's lk, from the staged rows
__device__ __forceinline__ void fin_position(const double *s_raw, const int *s_G, const double *eM, int tid, int p, int L,
                                             double lk, bool dead, double *res) {
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
}

// ONE entry (position p, row `row` of NUM_ROW) straight from the row sums in global memory: the expressions of fin_position,
// for the callers that want a handful of columns (pick_kernel: the candidate columns of a chunk).
__device__ __forceinline__ double fin_entry(const double *raw, const int *rawG, const double *eM, int p, uint32_t row, int L, double lk) {
    const bool dead = !(lk > JTK_LOG_ZERO);
    double res = JTK_LOG_ZERO - (dead ? 0.0 : lk);
    if (dead || p > L) return res;
    if (row < 4) {                       // sub: row p+1
        if (p + 1 <= L) {
            const double *a = raw + (uint64_t)(p + 1) * JTK_ACC_N;
            const uint32_t b = row;
            double v = eM[4 * b + 0] * a[0];
            v = fma(eM[4 * b + 1], a[1], v);
            v = fma(eM[4 * b + 2], a[2], v);
            v = fma(eM[4 * b + 3], a[3], v);
            v = v + a[4];
            res = fin_log(v, rawG[p + 1], lk);
        }
    } else if (row < 8) {                // ins: row p
        const double *a = raw + (uint64_t)p * JTK_ACC_N;
        const uint32_t b = row - 4;
        double v = eM[4 * b + 0] * a[5];
        v = fma(eM[4 * b + 1], a[6], v);
        v = fma(eM[4 * b + 2], a[7], v);
        v = fma(eM[4 * b + 3], a[8], v);
        v = v + a[9];
        res = fin_log(v, rawG[p], lk);
    } else if (row < 11) {               // copy c: row p+1
        if (p + 1 <= L) {
            const double *a = raw + (uint64_t)(p + 1) * JTK_ACC_N;
            res = fin_log(a[10 + (row - 8)], rawG[p + 1], lk);
        }
    } else {                             // del d: row p+d+1
        const int dd = (int)row - 10;
        if (p + dd + 1 <= L) {
            const double *a = raw + (uint64_t)(p + dd + 1) * JTK_ACC_N;
            res = fin_log(a[13 + dd - 1], rawG[p + dd + 1], lk);
        }
    }
    return res;
}

}  // namespace
