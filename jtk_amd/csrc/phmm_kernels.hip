// phmm_kernels.hip -- band preparation and table finalisation of the banded pair-HMM for gfx950 (CDNA4); the sweep itself
// (phmm_kernel) lives in phmm_sweep.hip.  (The round-2 sweep that used to sit here behind -DJTK_PHMM_WITH_R2 is in
// scripts/experiments/legacy/phmm_kernel_r2.hip.inc.)
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
#include "finalize_common.h"

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
// band_prep: the delta bits (c[t] - c[t-1]) of every read's band from its ops, the interval of anti-diagonals whose whole
// band lies inside the DP matrix, and the check that the ops consume exactly the template and the read.  A Match step
// visits c = i+1 on both of its diagonals.
// Round 6: one WAVE per read, 64 ops per step (rounds 1-5: one thread per read walking its ~2,300 ops with dependent 8-byte
// loads, 1.2 ms per polish round whatever the number of active reads).  An op's position (i, j) before it is a prefix count
// of the ops before it that consume a template / a read base: two ballots and two mbcnt per 64 ops; the delta bits are OR-ed
// into the wave's words in LDS and leave as whole words.  The walk is order-free: the interval is [first, last] diagonal that
// satisfies the condition (the set is one interval, see below), a min / max over the lanes.
// ------------------------------------------------------------------------------------------------------
#define BP_WAVES 4
__global__ __launch_bounds__(64 * BP_WAVES) void band_prep_kernel(uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                                                                 ChunkState *state, DevBufs bufs, uint64_t *delta, int only_active,
                                                                 uint32_t max_words) {
    extern __shared__ __align__(8) unsigned char bp_smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * BP_WAVES + wave;
    if (r >= n_reads) return;
    const ReadMeta rm = reads[r];
    ChunkState *st = &state[rm.chunk];
    if (st->status != 0) return;
    if (only_active && !st->active) return;
    unsigned long long *sd = reinterpret_cast<unsigned long long *>(bp_smem) + (size_t)wave * max_words;
    const uint32_t L = st->tmpl_len, n = rm.read_len, T = L + n;
    const uint8_t *ops = bufs.ops[st->buf] + rm.ops_off;
    const uint32_t n_ops = bufs.ops_len[st->buf][r];
    uint64_t *d = delta + rm.delta_off;
    const uint32_t words = (T >> 6) + 2;
    for (uint32_t w = lane; w < words; w += 64) sd[w] = 0ull;
    // the interval [f_lo, f_hi] of diagonals whose WHOLE band lies inside the DP matrix (0 <= i <= L, 0 <= j <= n for every
    // band cell): c - r >= 0, c + r <= L, c + r <= t, c - r >= t - n.  c and t - c are non-decreasing in t, so the set is one
    // interval; phmm_kernel runs it under an EXEC mask that is the band and takes per-lane predicates outside of it.
    const int32_t rad = (int32_t)chunks[rm.chunk].radius;
    uint32_t f_lo = 0xffffffffu, f_hi = 0;  // per lane: min / max of the diagonals it saw satisfy the condition
    auto visit = [&](uint32_t tt, uint32_t cc) {  // diagonal tt has centre cc
        const int32_t c = (int32_t)cc, td = (int32_t)tt;
        if (c - rad >= 0 && c + rad <= (int32_t)L && c + rad <= td && c - rad >= td - (int32_t)n) {
            f_lo = tt < f_lo ? tt : f_lo;
            f_hi = tt > f_hi ? tt : f_hi;
        }
    };
    uint32_t i0 = 0, j0 = 0;  // template / read bases consumed before this step's 64 ops (uniform)
    bool bad = false;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t base = 0; base < n_ops && !bad; base += 64) {
        const uint32_t k = base + lane;
        const bool has = k < n_ops;
        const uint32_t op = has ? ops[k] : (uint32_t)JTK_OP_INS;
        const bool invalid = has && op > JTK_OP_DEL;
        const bool di = has && (op <= JTK_OP_MISMATCH || op == JTK_OP_DEL), dj = has && (op <= JTK_OP_MISMATCH || op == JTK_OP_INS);
        const unsigned long long mi = __ballot(di), mj = __ballot(dj);
        const uint32_t i = i0 + (uint32_t)__popcll(mi & below), j = j0 + (uint32_t)__popcll(mj & below), t = i + j;  // before the op
        const uint32_t t_after = t + (di ? 1u : 0u) + (dj ? 1u : 0u);
        // (the serial walk stopped at the first op that passed diagonal T: nothing behind it left a bit or a visit; the read is
        // bad either way, and so is its chunk -- only the bounds matter here)
        if (has && !invalid && t_after <= T + 2) {
            if (op == JTK_OP_INS) {
                visit(t + 1, i);
            } else {
                if (t + 1 <= T) atomicOr(&sd[(t + 1) >> 6], 1ull << ((t + 1) & 63u));
                visit(t + 1, i + 1);
                if (op != JTK_OP_DEL) visit(t + 2, i + 1);
            }
        }
        if (__ballot(invalid || (has && t_after > T)) != 0ull) bad = true;
        i0 += (uint32_t)__popcll(mi);
        j0 += (uint32_t)__popcll(mj);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(f_lo, o, 64), b = __shfl_xor(f_hi, o, 64);
        f_lo = a < f_lo ? a : f_lo;
        f_hi = b > f_hi ? b : f_hi;
    }
    if (bad || i0 != L || j0 != n) {
        bad = true;
        if (lane == 0) atomicMin(&st->status, (int)JTK_ERR_OPS_MISMATCH);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the wave's own LDS atomics are done before its lanes read the words
    for (uint32_t w = lane; w < words; w += 64) d[w] = sd[w];
    // the word behind the read's delta words (session.hip sizes the slot from the template CAPACITY: one word to spare)
    if (lane == 0)
        d[((chunks[rm.chunk].tmpl_cap + rm.read_len) >> 6) + 2] = (bad || f_hi < f_lo) ? 1ull : ((uint64_t)f_hi << 32 | f_lo);
}


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
// (fin_log / fin_stage / fin_position / fin_entry: finalize_common.h -- shared with filter_kernels.hip)
__global__ __launch_bounds__(FIN_TILE) void finalize_kernel(uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                                                           const ChunkState *state, const HmmDev *hmm2, double *raw_all,
                                                           const int *rawG_all, const double *lk_all, int only_active,
                                                           int converged_in) {
    // A tile's raw rows come in through LDS with coalesced 16-byte loads and its 14 x 128 table entries leave the same way:
    // a thread's own rows (128 B apart) and its 112 bytes of output would otherwise be 16-byte pieces of different lines.
    __shared__ __align__(16) double s_raw[FIN_ROWS * FIN_PITCH];
    __shared__ int s_G[FIN_ROWS];
    const uint32_t item = blockIdx.x;
    const ReadMeta rm = reads[item];
    const ChunkState st = state[rm.chunk];
    if (st.status != 0) return;
    if (only_active && !st.active) return;
    // converged_in >= 0: only the chunks whose polishing converged in that round (select_edits_kernel found no edit: the
    // chunk left the active set with `rounds` = round + 1) -- their row sums become their final table, once
    if (converged_in >= 0 && (st.active || (int)st.rounds != converged_in + 1)) return;
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
        fin_stage(s_raw, s_G, raw, rawG, p0, n_rows, tid, dead);
        __syncthreads();
        double res[JTK_NUM_ROW];
        fin_position(s_raw, s_G, eM, tid, p, L, lk, dead, res);
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

// ------------------------------------------------------------------------------------------------------
// A polish round needs only the column totals  total[p][row] = sum over the voting reads, IN READ ORDER, of
// (table_r[p][row] - lk_r)  -- never the per-read tables of a template that is about to be edited.  This kernel is
// finalize_kernel's arithmetic (the same staged rows, the same 14 logs per position) with the ordered sum that was sum_tables_kernel (up to round 4) on top:
// one workgroup per (chunk, tile of 128 positions) walks the reads in order and keeps the 14 totals of its position in
// registers.  Per read and round it reads the 16 row sums per position once and writes nothing -- finalize + sum_tables wrote
// the 14-entry table in place and read it back (704 -> 256 bytes per position), and the row sums of a chunk stay intact, so
// a chunk that turns out to have converged gets its table from finalize_kernel afterwards (converged_in).  Bit for bit the
// totals of the two-kernel path: the same values are added in the same order.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FIN_TILE) void sum_final_kernel(const ReadMeta *reads, const ChunkMeta *chunks, const ChunkState *state,
                                                           const HmmDev *hmm2, const double *raw_all, const int *rawG_all,
                                                           const double *lk_all, double *total_all) {
    __shared__ __align__(16) double s_raw[FIN_ROWS * FIN_PITCH];
    __shared__ int s_G[FIN_ROWS];
    const uint32_t ci = blockIdx.y;
    const ChunkState st = state[ci];
    if (st.status != 0 || !st.active) return;
    const ChunkMeta cm = chunks[ci];
    const int L = (int)st.tmpl_len;
    const int p0 = (int)blockIdx.x * FIN_TILE;
    if (p0 > L) return;
    const int tid = threadIdx.x, p = p0 + tid;
    const int n_rows = min(FIN_ROWS, L + 1 - p0);
    double eMf[16], eMr[16];  // strand 1 -> hmm2[0], strand 0 -> hmm2[1] (as finalize_kernel)
#pragma unroll
    for (int k = 0; k < 16; k++) {
        eMf[k] = hmm2[0].eM[k];
        eMr[k] = hmm2[1].eM[k];
    }
    double tot[JTK_NUM_ROW];
#pragma unroll
    for (int k = 0; k < JTK_NUM_ROW; k++) tot[k] = 0.0;
    const uint32_t voters = (cm.take_num && cm.take_num < cm.n_reads) ? cm.take_num : cm.n_reads;
    for (uint32_t r = 0; r < voters; r++) {
        const uint32_t item = cm.read_first + r;
        const ReadMeta rm = reads[item];
        const double lk = lk_all[item];
        const bool dead = !(lk > JTK_LOG_ZERO);
        fin_stage(s_raw, s_G, raw_all + rm.raw_off, rawG_all + rm.row_off, p0, n_rows, tid, dead);
        __syncthreads();
        double res[JTK_NUM_ROW];
        if (rm.strand)
            fin_position(s_raw, s_G, eMf, tid, p, L, lk, dead, res);
        else
            fin_position(s_raw, s_G, eMr, tid, p, L, lk, dead, res);
#pragma unroll
        for (int k = 0; k < JTK_NUM_ROW; k++) tot[k] += res[k];
        __syncthreads();  // before the next read's rows are staged
    }
    if (p <= L) {
        double *total = total_all + cm.total_off + (uint64_t)p * JTK_NUM_ROW;
#pragma unroll
        for (int k = 0; k < JTK_NUM_ROW; k++) total[k] = tot[k];
    }
}

}  // namespace

void launch_band_prep(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                      ChunkState *state, DevBufs bufs, uint64_t *delta, int only_active, uint32_t max_tmpl, uint32_t max_read) {
    if (n_reads == 0) return;
    const uint32_t max_words = ((max_tmpl + max_read) >> 6) + 3;   // (T >> 6) + 2 words of the longest read, T <= max_tmpl + max_read
    band_prep_kernel<<<(n_reads + BP_WAVES - 1) / BP_WAVES, 64 * BP_WAVES, (size_t)BP_WAVES * max_words * 8, s>>>(
        n_reads, reads, chunks, state, bufs, delta, only_active, max_words);
}


void launch_finalize(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                     const ChunkState *state, const HmmDev *hmm2, double *raw, const int *rawG, const double *lk,
                     uint32_t max_tmpl, int only_active, int converged_in) {
    if (n_reads == 0) return;
    (void)max_tmpl;
    finalize_kernel<<<n_reads, FIN_TILE, 0, s>>>(n_reads, reads, chunks, state, hmm2, raw, rawG, lk, only_active, converged_in);
}

// the column totals of a polish round straight from the row sums (sum_final_kernel): total[chunk][p][row]
void launch_sum_final(hipStream_t s, uint32_t n_chunks, const ReadMeta *reads, const ChunkMeta *chunks, const ChunkState *state,
                      const HmmDev *hmm2, const double *raw, const int *rawG, const double *lk, double *total, uint32_t max_tmpl) {
    if (n_chunks == 0) return;
    dim3 grid((max_tmpl + 1 + FIN_TILE - 1) / FIN_TILE, n_chunks);
    sum_final_kernel<<<grid, FIN_TILE, 0, s>>>(reads, chunks, state, hmm2, raw, rawG, lk, total);
}
