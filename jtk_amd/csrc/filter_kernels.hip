// filter_kernels.hip -- per-site likelihood-gain scoring and variant-column selection on the device.
//
// Follows haplotyper/src/local_clustering/pseudo_mcmc.rs:
//   compress_small_gains :141-165   column_sum :577-588        filter_profiles :426-474
//   is_in_short_homopolymer :497-514   has_small_pvalue :476-495   is_explainable_by_strandedness :314-339
//   poisson_lk :636-638   pick_filtered_profiles :516-575 (find_next_variants :590-600,
//   cosine_similarity :602-615, sokal_michener :618-633)   filter_by :70-75
//   operation_and_homopolymer_length :180-193
// and haplotyper/src/likelihood_gains.rs: Gains::expected :79-87, pvalues :115-129, Pvalues::pvalue :149-158.
//
// The N x 14(L+1) table (already minus lk) is read column-wise: thread = column, loop over the reads in
// read order, so every sum has the reference's left-to-right order and neighbouring threads read
// neighbouring addresses.  compress_small_gains is applied on the fly (the table itself stays
// uncompressed because polishing needs it).
#include "device_common.h"
#include "finalize_common.h"

namespace {

__device__ __forceinline__ int diff_type_of(uint32_t row) {  // pos_to_bp_and_difftype :168-178
    return row < 4 ? JTK_DIFF_SUBST : (row < 8 + JTK_COPY_SIZE ? JTK_DIFF_INS : JTK_DIFF_DEL);
}
__device__ __forceinline__ double gains_expected(const jtk_gains_t *g, uint32_t homop_len, int dt) {
    if (homop_len == 0) homop_len = 1;
    const uint32_t h = homop_len < g->max_homopolymer_len ? homop_len : g->max_homopolymer_len;
    return dt == JTK_DIFF_SUBST ? g->subst[h - 1].gain
                                : (dt == JTK_DIFF_DEL ? g->deletions[h - 1].gain : g->insertions[h - 1].gain);
}
__device__ __forceinline__ double compress(double x, double min_req) { return fabs(x) < min_req ? 0.0 : x; }

// homopolymer_length (:195-211): homop[p] = length of the run containing p
__global__ void homop_kernel(const ChunkMeta *chunks, const ChunkState *state, DevBufs bufs, uint16_t *homop_all,
                             const uint64_t *homop_off) {
    const uint32_t ci = blockIdx.y;
    const ChunkState st = state[ci];
    if (st.status != 0) return;
    const ChunkMeta cm = chunks[ci];
    const uint32_t L = st.tmpl_len, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= L) return;
    const uint8_t *x = bufs.tmpl[st.buf] + cm.tmpl_off;
    const uint8_t b = x[p];
    uint32_t a = p, e = p;
    while (a > 0 && x[a - 1] == b) a--;
    while (e + 1 < L && x[e + 1] == b) e++;
    homop_all[homop_off[ci] + p] = (uint16_t)(e - a + 1);
}

// Per chunk tables: pvalues for the 3 x H gain profiles at n = n_reads (likelihood_gains.rs:115-129),
// ln-factorials and ln(coverage * k).  Layout of aux (doubles) per chunk:
//   [0 .. 3*H*(n+1))  pv[type][h][count]   then lfact[0..n]   then lnlam[1..copy_num] at [.. + k]
__global__ void chunk_tables_kernel(const ChunkMeta *chunks, const ChunkState *state, const jtk_lc_params_t *params,
                                    double *aux_all, const uint64_t *aux_off) {
    const uint32_t ci = blockIdx.x;
    if (state[ci].status != 0) return;
    const ChunkMeta cm = chunks[ci];
    const jtk_gains_t *g = &params->gains;
    const uint32_t n = cm.n_reads, H = g->max_homopolymer_len;
    double *aux = aux_all + aux_off[ci];
    const uint32_t tid = threadIdx.x;
    if (tid < 3 * H) {
        const uint32_t type = tid / H, h = tid % H;
        const double prob = type == JTK_DIFF_SUBST ? g->subst[h].prob
                                                   : (type == JTK_DIFF_DEL ? g->deletions[h].prob : g->insertions[h].prob);
        double *out = aux + (uint64_t)tid * (n + 1);
        const double ln = jtk_log(prob), in_ln = jtk_log(1.0 - prob);
        out[0] = in_ln * (double)n;
        for (uint32_t k = 0; k < n; k++) {
            const double offset = ln + jtk_log((double)(n - k)) - in_ln - jtk_log((double)(k + 1));
            out[k + 1] = out[k] + offset;
        }
        for (uint32_t k = n; k-- > 0;) {
            const double x = out[k + 1], y = out[k];
            out[k] = y < x ? x + jtk_log(1.0 + jtk_exp(y - x)) : y + jtk_log(1.0 + jtk_exp(x - y));
        }
        for (uint32_t k = 0; k <= n; k++) out[k] = jtk_exp(out[k]);
    } else if (tid == 3 * H) {
        double *lfact = aux + (uint64_t)3 * H * (n + 1);
        double s = 0.0;
        lfact[0] = 0.0;
        for (uint32_t c = 1; c <= n; c++) {
            s += jtk_log((double)c);
            lfact[c] = s;
        }
    } else if (tid == 3 * H + 1) {
        double *lnlam = aux + (uint64_t)3 * H * (n + 1) + (n + 1);
        for (uint32_t k = 1; k <= cm.copy_num; k++) lnlam[k] = jtk_log(params->haploid_coverage * (double)k);
    }
}

// column statistics + every per-column filter of filter_profiles; cand[col] = total_lk (> 0) or -1.
__global__ void column_filter_kernel(const ReadMeta *reads, const ChunkMeta *chunks, const ChunkState *state,
                                     DevBufs bufs, const jtk_lc_params_t *params, const double *table_all,
                                     const uint16_t *homop_all, const uint64_t *homop_off, const double *aux_all,
                                     const uint64_t *aux_off, double *cand_all) {
    const uint32_t ci = blockIdx.y;
    const ChunkState st = state[ci];
    if (st.status != 0) return;
    const ChunkMeta cm = chunks[ci];
    const uint32_t L = st.tmpl_len, cols = JTK_NUM_ROW * (L + 1);
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= cols) return;
    double *cand = cand_all + cm.cand_off;
    cand[col] = -1.0;
    if (cm.copy_num < 2) return;  // clustering() returns before search_variants (:86-88)
    const uint32_t bp = col / JTK_NUM_ROW, row = col % JTK_NUM_ROW, temp_len = L + 1;
    if (!(JTK_MASK_LENGTH <= bp && bp + JTK_MASK_LENGTH <= temp_len)) return;
    if (!(row < 8 || row == 8 + JTK_COPY_SIZE)) return;
    const uint8_t *x = bufs.tmpl[st.buf] + cm.tmpl_off;
    const uint16_t *homop = homop_all + homop_off[ci];
    const int dt = diff_type_of(row);
    // is_in_short_homopolymer
    if (dt == JTK_DIFF_INS) {
        const uint8_t base = (uint8_t)(row - 4);  // rows 4..7 only reach here
        const uint32_t prev_len = bp > 0 ? homop[bp - 1] + (x[bp - 1] == base) : 0;
        const uint32_t next_len = bp < L ? homop[bp] + (x[bp] == base) : 0;
        if (!(prev_len <= JTK_MAX_HOMOP_LENGTH && next_len <= JTK_MAX_HOMOP_LENGTH)) return;
    } else if (dt == JTK_DIFF_DEL && bp < L) {
        if (!(homop[bp] <= JTK_MAX_HOMOP_LENGTH)) return;
    }
    const jtk_gains_t *g = &params->gains;
    const uint32_t n = cm.n_reads;
    const double min_req = gains_expected(g, bp < L ? homop[bp] : 1, dt) * 0.5;  // MIN_REQ_FRACTION
    // column_sum over compressed profiles
    double gain = 0.0;
    uint32_t count = 0;
    for (uint32_t r = 0; r < n; r++) {
        const double v = compress(table_all[reads[cm.read_first + r].table_off + col], min_req);
        if (JTK_POS_THR < v) {
            gain += v;
            count++;
        }
    }
    // has_small_pvalue
    const uint32_t H = g->max_homopolymer_len;
    const double *aux = aux_all + aux_off[ci];
    {
        const uint32_t homop_len = bp < L ? homop[bp] : 0;
        const uint32_t hh = (homop_len < H ? homop_len : H);
        const double pv = aux[((uint64_t)dt * H + (hh - 1)) * (n + 1) + count];
        const double expt = gains_expected(g, homop_len, dt) * 0.8;  // EXPT_GAIN_FACTOR
        const double pvalue = (double)temp_len * pv;
        if (!((double)count * expt < gain && pvalue < 0.05 / (double)temp_len)) return;
    }
    // is_explainable_by_strandedness
    {
        uint32_t strand_count[2] = {0, 0}, sign_count[2] = {0, 0}, obs[2][2] = {{0, 0}, {0, 0}};
        for (uint32_t r = 0; r < n; r++) {
            const ReadMeta &rm = reads[cm.read_first + r];
            const double v = compress(table_all[rm.table_off + col], min_req);
            if (!(fabs(v) > 0.0001)) continue;
            const uint32_t s = rm.strand ? 1 : 0;
            const uint32_t sg = (jtk_f64_bits(v) >> 63) ? 0 : 1;
            strand_count[s]++;
            sign_count[sg]++;
            obs[s][sg]++;
        }
        const uint32_t sum = strand_count[0] + strand_count[1];
        if (sum == 0) return;
        double chisq = 0.0;
        for (int s = 0; s < 2; s++) {
            double inner = 0.0;
            for (int sg = 0; sg < 2; sg++) {
                const double expected = (double)(strand_count[s] * sign_count[sg]) / (double)sum;
                const double d = (double)obs[s][sg] - expected;
                inner += d * d / expected;  // 0/0 = NaN when a strand or a sign is absent, as in the reference
            }
            chisq += inner;
        }
        if (!(chisq < 10.0)) return;
    }
    // total_lk = max_k poisson_lk(count, coverage*k) + gain
    const double *lfact = aux + (uint64_t)3 * H * (n + 1);
    const double *lnlam = lfact + (n + 1);
    double max_lk = 0.0;
    for (uint32_t k = 1; k <= cm.copy_num; k++) {
        const double lam = params->haploid_coverage * (double)k;
        const double v = (double)count * lnlam[k] - lam - lfact[count];
        if (k == 1 || !(v < max_lk)) max_lk = v;
    }
    const double total_lk = max_lk + gain;
    if (0.0 < total_lk) cand[col] = total_lk;
}

// ------------------------------------------------------------------------------------------------------
// column_filter_fused_kernel (round 6): the same statistics and filters as column_filter_kernel, taken STRAIGHT FROM THE ROW
// SUMS -- the N x 14(L+1) table is never written (`table - lk -> compress_small_gains -> column
// (sum, count)` as an epilogue; until round 5 finalize_kernel wrote every read's table in place, 224 KB per read, and this
// filter read it back column by column).  One workgroup per (chunk, tile of 128 positions) walks the reads IN READ ORDER like
// sum_final_kernel: a read's rows of the tile are staged in LDS, thread p evaluates the 14 entries of its position with
// finalize's own expressions (finalize_common.h: the same bits) and feeds the nine rows filter_profiles looks at (rows 0-7 and
// 11, pseudo_mcmc.rs:447-449) into per-column accumulators: the gain and count of column_sum (:577-588, compressed values above
// POS_THR, left to right over the reads) and the 2 x 2 strand / sign table of is_explainable_by_strandedness (:314-339).
// The per-column tests follow when the last read is through.
// ------------------------------------------------------------------------------------------------------
#define FF_ROWS 9  // rows 0..7 (substitutions, insertions) and 11 (1-bp deletion)
__device__ __forceinline__ uint32_t ff_row(int j) { return j < 8 ? (uint32_t)j : 8u + JTK_COPY_SIZE; }
__global__ __launch_bounds__(FIN_TILE) void column_filter_fused_kernel(const ReadMeta *reads, const ChunkMeta *chunks,
                                                                      const ChunkState *state, DevBufs bufs,
                                                                      const jtk_lc_params_t *params, const HmmDev *hmm2,
                                                                      const double *raw_all, const int *rawG_all,
                                                                      const double *lk_all, const uint16_t *homop_all,
                                                                      const uint64_t *homop_off, const double *aux_all,
                                                                      const uint64_t *aux_off, double *cand_all) {
    __shared__ __align__(16) double s_raw[FIN_ROWS * FIN_PITCH];
    __shared__ int s_G[FIN_ROWS];
    const uint32_t ci = blockIdx.y;
    const ChunkState st = state[ci];
    if (st.status != 0) return;
    const ChunkMeta cm = chunks[ci];
    const int L = (int)st.tmpl_len;
    const int p0 = (int)blockIdx.x * FIN_TILE;
    if (p0 > L) return;
    const int tid = threadIdx.x, p = p0 + tid;
    const uint32_t temp_len = (uint32_t)L + 1;
    double *cand = cand_all + cm.cand_off;
    if (p <= L) {
#pragma unroll
        for (uint32_t row = 0; row < JTK_NUM_ROW; row++) cand[(uint32_t)p * JTK_NUM_ROW + row] = -1.0;
    }
    if (cm.copy_num < 2) return;  // clustering() returns before search_variants (:86-88)
    const uint8_t *x = bufs.tmpl[st.buf] + cm.tmpl_off;
    const uint16_t *homop = homop_all + homop_off[ci];
    const jtk_gains_t *g = &params->gains;
    const uint32_t n = cm.n_reads, bp = (uint32_t)p;
    // which of the nine columns of this position survive the position filters (:447-456, is_in_short_homopolymer :497-514)
    uint32_t live = 0;
    if (p <= L && JTK_MASK_LENGTH <= bp && bp + JTK_MASK_LENGTH <= temp_len) {
#pragma unroll
        for (int j = 0; j < FF_ROWS; j++) {
            const uint32_t row = ff_row(j);
            const int dt = diff_type_of(row);
            bool ok = true;
            if (dt == JTK_DIFF_INS) {
                const uint8_t base = (uint8_t)(row - 4);
                const uint32_t prev_len = bp > 0 ? homop[bp - 1] + (x[bp - 1] == base) : 0;
                const uint32_t next_len = bp < (uint32_t)L ? homop[bp] + (x[bp] == base) : 0;
                ok = prev_len <= JTK_MAX_HOMOP_LENGTH && next_len <= JTK_MAX_HOMOP_LENGTH;
            } else if (dt == JTK_DIFF_DEL && bp < (uint32_t)L) {
                ok = homop[bp] <= JTK_MAX_HOMOP_LENGTH;
            }
            if (ok) live |= 1u << j;
        }
    }
    if (!__syncthreads_or(live != 0)) return;  // nothing of this tile can become a candidate
    const uint32_t homop_here = (p <= L && bp < (uint32_t)L) ? homop[bp] : 1;
    // MIN_REQ_FRACTION (:141-165): one value per difference type at this position
    const double mr_sub = gains_expected(g, homop_here, JTK_DIFF_SUBST) * 0.5, mr_ins = gains_expected(g, homop_here, JTK_DIFF_INS) * 0.5,
                 mr_del = gains_expected(g, homop_here, JTK_DIFF_DEL) * 0.5;
    double gain[FF_ROWS];
    // count | obs[strand][sign] as 16-bit fields (launch_filter takes this kernel for pile-ups below 65,536 reads):
    // cnt_neg0[j] = count | obs[0][0] << 16, pos[j] = obs[0][1] | obs[1][1] << 16, neg1[j] = obs[1][0]
    uint32_t cnt_neg0[FF_ROWS], pos[FF_ROWS], neg1[FF_ROWS];
#pragma unroll
    for (int j = 0; j < FF_ROWS; j++) {
        gain[j] = 0.0;
        cnt_neg0[j] = pos[j] = neg1[j] = 0;
    }
    const int n_rows = min(FIN_ROWS, L + 1 - p0);
    for (uint32_t r = 0; r < n; r++) {
        const uint32_t item = cm.read_first + r;
        const ReadMeta rm = reads[item];
        const double lk = lk_all[item];
        const bool dead = !(lk > JTK_LOG_ZERO);
        fin_stage(s_raw, s_G, raw_all + rm.raw_off, rawG_all + rm.row_off, p0, n_rows, tid, dead);
        __syncthreads();
        double res[JTK_NUM_ROW];
        double eM[16];  // strand 1 -> hmm2[0], strand 0 -> hmm2[1] (as finalize_kernel); uniform: scalar loads, once per read
        {
            const double *src = rm.strand ? hmm2[0].eM : hmm2[1].eM;
#pragma unroll
            for (int k = 0; k < 16; k++) eM[k] = src[k];
        }
        fin_position(s_raw, s_G, eM, tid, p, L, lk, dead, res);
        const uint32_t strand = rm.strand ? 1u : 0u;
#pragma unroll
        for (int j = 0; j < FF_ROWS; j++) {
            const double v = compress(res[ff_row(j)], j < 4 ? mr_sub : (j < 8 ? mr_ins : mr_del));
            if (JTK_POS_THR < v) {
                gain[j] += v;
                cnt_neg0[j] += 1u;
            }
            if (fabs(v) > 0.0001) {
                if (jtk_f64_bits(v) >> 63) {   // negative
                    if (strand) neg1[j] += 1u; else cnt_neg0[j] += 1u << 16;
                } else {
                    pos[j] += strand ? 1u << 16 : 1u;
                }
            }
        }
        __syncthreads();  // before the next read's rows are staged
    }
    if (live == 0) return;
    const uint32_t H = g->max_homopolymer_len;
    const double *aux = aux_all + aux_off[ci];
    const double *lfact = aux + (uint64_t)3 * H * (n + 1);
    const double *lnlam = lfact + (n + 1);
#pragma unroll
    for (int j = 0; j < FF_ROWS; j++) {
        if (!((live >> j) & 1u)) continue;
        const uint32_t row = ff_row(j), col = bp * JTK_NUM_ROW + row;
        const uint32_t count_j = cnt_neg0[j] & 0xffffu, o00 = cnt_neg0[j] >> 16, o01 = pos[j] & 0xffffu, o11 = pos[j] >> 16, o10 = neg1[j];
        const int dt = diff_type_of(row);
        // has_small_pvalue (:476-495)
        {
            const uint32_t homop_len = bp < (uint32_t)L ? homop[bp] : 0;
            const uint32_t hh = (homop_len < H ? homop_len : H);
            const double pv = aux[((uint64_t)dt * H + (hh - 1)) * (n + 1) + count_j];
            const double expt = gains_expected(g, homop_len, dt) * 0.8;  // EXPT_GAIN_FACTOR
            const double pvalue = (double)temp_len * pv;
            if (!((double)count_j * expt < gain[j] && pvalue < 0.05 / (double)temp_len)) continue;
        }
        // is_explainable_by_strandedness (:314-339)
        {
            const uint32_t obs[2][2] = {{o00, o01}, {o10, o11}};
            const uint32_t strand_count[2] = {o00 + o01, o10 + o11}, sign_count[2] = {o00 + o10, o01 + o11};
            const uint32_t sum = strand_count[0] + strand_count[1];
            if (sum == 0) continue;
            double chisq = 0.0;
            for (int s = 0; s < 2; s++) {
                double inner = 0.0;
                for (int sg = 0; sg < 2; sg++) {
                    const double expected = (double)(strand_count[s] * sign_count[sg]) / (double)sum;
                    const double d = (double)obs[s][sg] - expected;
                    inner += d * d / expected;  // 0/0 = NaN when a strand or a sign is absent, as in the reference
                }
                chisq += inner;
            }
            if (!(chisq < 10.0)) continue;
        }
        // total_lk = max_k poisson_lk(count, coverage*k) + gain (:457-461, :636-638)
        double max_lk = 0.0;
        for (uint32_t k = 1; k <= cm.copy_num; k++) {
            const double lam = params->haploid_coverage * (double)k;
            const double v = (double)count_j * lnlam[k] - lam - lfact[count_j];
            if (k == 1 || !(v < max_lk)) max_lk = v;
        }
        const double total_lk = max_lk + gain[j];
        if (0.0 < total_lk) cand[col] = total_lk;
    }
}

// one wave per chunk: ordered compaction of the candidates, the greedy pick, then the feature matrix.
// TABLE = true: the entries come from the materialised table (finalize_kernel ran: rounds 1-5, JTK_FILTER_FUSED=0); false: every
// entry of a candidate column is evaluated from the row sums when it is needed (fin_entry: the same expressions, the same bits)
// TRACE = true (pick_trace_kernel, jtk_lc_session_trace): the same pick once more for ONE chunk, leaving what the reference's
// trace! rows need -- tr[0] = candidates (TOTAL :467), tr[1] = picks, tr[2 ..] = the picked candidates' list indices in pick order
// (PICK :539), tr_count[i] = reads with a gain above POS_THR in candidate i's column (CAND :471, column_sum's count :577-588).
template <bool TABLE, bool TRACE>
__device__ __forceinline__ void pick_body(const uint32_t ci, const ReadMeta *reads, const ChunkMeta *chunks, ChunkState *state,
                                          const jtk_lc_params_t *params, const double *table_all, const HmmDev *hmm2,
                                          const int *rawG_all, const double *lk_all,
                                          const uint16_t *homop_all, const uint64_t *homop_off,
                                          const double *cand_all, uint32_t *list_all, uint8_t *sel_all,
                                          double *feat_all, uint32_t *vtype_all, uint32_t *pos_all, uint32_t *tr, uint32_t *tr_count) {
    ChunkState *st = &state[ci];
    if (st->status != 0) return;
    const ChunkMeta cm = chunks[ci];
    const uint32_t lane = threadIdx.x;
    const uint32_t L = st->tmpl_len, cols = JTK_NUM_ROW * (L + 1), n = cm.n_reads;
    const double *cand = cand_all + cm.cand_off;
    uint32_t *list = list_all + cm.cand_off;  // candidate columns in ascending order
    uint8_t *sel = sel_all + cm.cand_off;
    const uint16_t *homop = homop_all + homop_off[ci];
    const jtk_gains_t *g = &params->gains;
    auto entry = [&](uint32_t r, uint32_t col) -> double {  // table_r[col] (already minus lk_r)
        const uint32_t item = cm.read_first + r;
        const ReadMeta &rm = reads[item];
        if (TABLE) return table_all[rm.table_off + col];
        return fin_entry(table_all + rm.raw_off, rawG_all + rm.row_off, (rm.strand ? hmm2[0] : hmm2[1]).eM, (int)(col / JTK_NUM_ROW),
                         col % JTK_NUM_ROW, (int)L, lk_all[item]);
    };
    __shared__ uint32_t s_np;
    // ---- ordered compaction
    uint32_t np = 0;
    for (uint32_t base = 0; base < cols; base += 64) {
        const uint32_t col = base + lane;
        const bool is = col < cols && cand[col] > 0.0;
        const unsigned long long m = __ballot(is);
        if (is) {
            const uint32_t before = __popcll(m & ((1ull << lane) - 1ull));
            list[np + before] = col;
            sel[np + before] = 0;
        }
        np += __popcll(m);
    }
    __syncthreads();
    auto min_req_of = [&](uint32_t col) {
        const uint32_t bp = col / JTK_NUM_ROW, row = col % JTK_NUM_ROW;
        return gains_expected(g, bp < L ? homop[bp] : 1, diff_type_of(row)) * 0.5;
    };
    uint32_t n_picks = 0;
    if (TRACE) {
        for (uint32_t i = lane; i < np; i += 64) {
            const uint32_t col = list[i];
            const double mr = min_req_of(col);
            uint32_t cnt = 0;
            for (uint32_t r = 0; r < n; r++)
                if (JTK_POS_THR < compress(entry(r, col), mr)) cnt++;
            tr_count[i] = cnt;
        }
        if (lane == 0) tr[0] = np;
    }
    // ---- pick_filtered_profiles
    const uint32_t per_round = cm.copy_num > 2 ? cm.copy_num : 2;
    for (uint32_t round = 0; round < 3; round++) {
        for (uint32_t i = lane; i < np; i += 64)
            if (sel[i] == 3) sel[i] = 0;
        __syncthreads();
        for (uint32_t it = 0; it < per_round; it++) {
            // find_next_variants: LAST maximum among flag == 0
            double bv = -1.0;
            int bi = -1;
            for (uint32_t i = lane; i < np; i += 64)
                if (sel[i] == 0) {
                    const double v = cand[list[i]];
                    if (bi < 0 || !(v < bv)) {
                        bv = v;
                        bi = (int)i;
                    }
                }
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi > bi))) {
                    bv = ov;
                    bi = oi;
                }
            }
            if (bi < 0) break;
            const uint32_t picked = list[bi], picked_bp = picked / JTK_NUM_ROW;
            const double mr_p = min_req_of(picked);
            __syncthreads();
            if (lane == 0) sel[bi] = 1;
            if (TRACE) {
                if (lane == 0 && n_picks < JTK_TRACE_MAX_PICKS) tr[2 + n_picks] = (uint32_t)bi;
                n_picks++;
            }
            __syncthreads();
            for (uint32_t i = lane; i < np; i += 64) {
                if (!(sel[i] == 0 || sel[i] == 3)) continue;
                const uint32_t col = list[i], bp = col / JTK_NUM_ROW;
                const uint32_t diff = bp > picked_bp ? bp - picked_bp : picked_bp - bp;
                if (diff < JTK_MASK_LENGTH) {
                    sel[i] = 2;
                    continue;
                }
                const double mr_i = min_req_of(col);
                uint32_t mat = 0, mism = 0;
                double ip = 0.0, isq = 0.0, jsq = 0.0;
                for (uint32_t r = 0; r < n; r++) {
                    const double xv = compress(entry(r, picked), mr_p), yv = compress(entry(r, col), mr_i);
                    if (JTK_POS_THR < fabs(xv) && JTK_POS_THR < fabs(yv)) {
                        if (0.0 < xv * yv)
                            mat++;
                        else
                            mism++;
                        ip = ip + xv * yv;
                        isq = isq + xv * xv;
                        jsq = jsq + yv * yv;
                    }
                }
                const uint32_t tot = mat + mism;
                const double sok = tot == 0 ? 0.0 : (double)(mism > mat ? mism : mat) / (double)tot;
                const double cs = isq == 0.0 ? 0.0 : ip / sqrt(isq) / sqrt(jsq);
                if (0.8 < sok || 0.8 < fabs(cs)) sel[i] = 3;
            }
            __syncthreads();
        }
    }
    if (TRACE && lane == 0) tr[1] = n_picks < JTK_TRACE_MAX_PICKS ? n_picks : JTK_TRACE_MAX_PICKS;
    // ---- selected probes in candidate order; features (filter_by) and variant types
    if (lane == 0) {
        uint32_t d = 0;
        for (uint32_t i = 0; i < np && d < JTK_MAX_DIM; i++)
            if (sel[i] == 1) {
                const uint32_t col = list[i], bp = col / JTK_NUM_ROW;
                pos_all[(uint64_t)ci * JTK_MAX_DIM + d] = col;
                vtype_all[2 * ((uint64_t)ci * JTK_MAX_DIM + d)] = bp < L ? homop[bp] : 0;
                vtype_all[2 * ((uint64_t)ci * JTK_MAX_DIM + d) + 1] = (uint32_t)diff_type_of(col % JTK_NUM_ROW);
                d++;
            }
        s_np = d;
        st->dim = d;
    }
    __syncthreads();
    const uint32_t D = s_np;
    double *feat = feat_all + cm.feat_off;
    for (uint32_t e = lane; e < n * D; e += 64) {
        const uint32_t r = e / D, d = e % D;
        const uint32_t col = pos_all[(uint64_t)ci * JTK_MAX_DIM + d];
        feat[(uint64_t)r * D + d] = compress(entry(r, col), min_req_of(col));
    }
}

template <bool TABLE>
__global__ __launch_bounds__(64) void pick_kernel(const ReadMeta *reads, const ChunkMeta *chunks, ChunkState *state,
                                                  const jtk_lc_params_t *params, const double *table_all, const HmmDev *hmm2,
                                                  const int *rawG_all, const double *lk_all,
                                                  const uint16_t *homop_all, const uint64_t *homop_off,
                                                  const double *cand_all, uint32_t *list_all, uint8_t *sel_all,
                                                  double *feat_all, uint32_t *vtype_all, uint32_t *pos_all) {
    pick_body<TABLE, false>(blockIdx.x, reads, chunks, state, params, table_all, hmm2, rawG_all, lk_all, homop_all, homop_off, cand_all,
                            list_all, sel_all, feat_all, vtype_all, pos_all, nullptr, nullptr);
}
template <bool TABLE>
__global__ __launch_bounds__(64) void pick_trace_kernel(uint32_t ci, const ReadMeta *reads, const ChunkMeta *chunks, ChunkState *state,
                                                        const jtk_lc_params_t *params, const double *table_all, const HmmDev *hmm2,
                                                        const int *rawG_all, const double *lk_all,
                                                        const uint16_t *homop_all, const uint64_t *homop_off,
                                                        const double *cand_all, uint32_t *list_all, uint8_t *sel_all,
                                                        double *feat_all, uint32_t *vtype_all, uint32_t *pos_all, uint32_t *tr,
                                                        uint32_t *tr_count) {
    pick_body<TABLE, true>(ci, reads, chunks, state, params, table_all, hmm2, rawG_all, lk_all, homop_all, homop_off, cand_all, list_all,
                           sel_all, feat_all, vtype_all, pos_all, tr, tr_count);
}

}  // namespace

// jtk_lc_session_trace: chunk `ci`'s pick again (the candidates, list and every output come out as they were) with the trace
// arrays filled: tr = 2 + JTK_TRACE_MAX_PICKS words, tr_count = one word per candidate (<= the chunk's column count).
void launch_pick_trace(hipStream_t s, uint32_t ci, const ReadMeta *reads, const ChunkMeta *chunks, ChunkState *state,
                       const jtk_lc_params_t *params, const double *table, const uint16_t *homop, const uint64_t *homop_off,
                       const double *cand, uint32_t *list, uint8_t *sel, double *feat, uint32_t *vtype, uint32_t *pos,
                       const HmmDev *hmm2, const int *rawG, const double *lk, int fused, uint32_t *tr, uint32_t *tr_count) {
    if (fused)
        pick_trace_kernel<false><<<1, 64, 0, s>>>(ci, reads, chunks, state, params, table, hmm2, rawG, lk, homop, homop_off, cand, list,
                                                  sel, feat, vtype, pos, tr, tr_count);
    else
        pick_trace_kernel<true><<<1, 64, 0, s>>>(ci, reads, chunks, state, params, table, hmm2, rawG, lk, homop, homop_off, cand, list,
                                                 sel, feat, vtype, pos, tr, tr_count);
}

void launch_filter(hipStream_t s, uint32_t n_chunks, const ReadMeta *reads, const ChunkMeta *chunks,
                   ChunkState *state, DevBufs bufs, const jtk_lc_params_t *params, const double *table,
                   uint16_t *homop, const uint64_t *homop_off, double *aux, const uint64_t *aux_off, double *cand,
                   uint32_t *list, uint8_t *sel, double *feat, uint32_t *vtype, uint32_t *pos, uint32_t max_tmpl,
                   const HmmDev *hmm2, const int *rawG, const double *lk, int fused) {
    if (n_chunks == 0) return;
    {
        dim3 grid((max_tmpl + 127) / 128, n_chunks);
        homop_kernel<<<grid, 128, 0, s>>>(chunks, state, bufs, homop, homop_off);
    }
    chunk_tables_kernel<<<n_chunks, 64, 0, s>>>(chunks, state, params, aux, aux_off);
    if (fused) {  // `table` holds the raw row sums (finalize_kernel did not run): statistics and picks straight from them
        dim3 grid((max_tmpl + 1 + FIN_TILE - 1) / FIN_TILE, n_chunks);
        column_filter_fused_kernel<<<grid, FIN_TILE, 0, s>>>(reads, chunks, state, bufs, params, hmm2, table, rawG, lk, homop, homop_off,
                                                            aux, aux_off, cand);
        pick_kernel<false><<<n_chunks, 64, 0, s>>>(reads, chunks, state, params, table, hmm2, rawG, lk, homop, homop_off, cand, list,
                                                   sel, feat, vtype, pos);
        return;
    }
    {
        const uint32_t cols = JTK_NUM_ROW * (max_tmpl + 1);
        dim3 grid((cols + 127) / 128, n_chunks);
        column_filter_kernel<<<grid, 128, 0, s>>>(reads, chunks, state, bufs, params, table, homop, homop_off, aux,
                                                   aux_off, cand);
    }
    pick_kernel<true><<<n_chunks, 64, 0, s>>>(reads, chunks, state, params, table, hmm2, rawG, lk, homop, homop_off, cand, list, sel,
                                              feat, vtype, pos);
}
