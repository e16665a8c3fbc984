// polish_kernels.hip -- one round of consensus polishing on the device (own specification of kiley
// `polish_until_converge_antidiagonal`, local_clustering/mod.rs:105-106; see DESIGN.md and oracle/phmm.c).
//
//   total[p][row] = sum over the reads of the pile-up, IN READ ORDER, of (table_r - lk_r)      (sum_final_kernel, phmm_kernels.hip)
//   scan p = ignore_edge .. L-ignore_edge-1 left to right: first best row; apply if > MIN_GAIN, then skip
//   the touched bases + inactive(round) positions; build the edited template                    (select_edits)
//   re-thread every read's ops around the edits and recompute the Match/Mismatch tags           (rethread)
//   flip the ping-pong buffers                                                                  (commit)
#include "device_common.h"

namespace {

__device__ __forceinline__ uint32_t edit_inserted(uint32_t row, uint32_t pos, uint32_t L) {
    if (row >= 4 && row < 8) return 1;
    if (row >= 8 && row < 11) {
        const uint32_t c = row - 7;
        return pos + c <= L ? c : L - pos;
    }
    return 0;
}

// one wave per chunk: lanes reduce the 14 rows of each position in parallel; lane 0 does the (inherently
// sequential) left-to-right scan and writes the edited template.
__global__ __launch_bounds__(64) void select_edits_kernel(const ChunkMeta *chunks, ChunkState *state, DevBufs bufs,
                                                          const double *total_all, Edit *edits_all,
                                                          uint32_t *new_len, uint32_t ignore_edge, int final_pass) {
    extern __shared__ unsigned char s_best[];  // per position: best row | 0x80 if its total > MIN_GAIN
    const uint32_t ci = blockIdx.x;
    ChunkState *st = &state[ci];
    if (st->status != 0 || !st->active) return;
    const ChunkMeta cm = chunks[ci];
    const uint32_t L = st->tmpl_len;
    if (final_pass) {  // table of the final template is ready; nothing more to select
        if (threadIdx.x == 0) {
            st->active = 0;
            st->n_edits = 0;
        }
        return;
    }
    const double *total = total_all + cm.total_off;
    for (uint32_t p = threadIdx.x; p < L; p += 64) {
        uint32_t best = 0;
        double g = total[p * JTK_NUM_ROW];
        for (uint32_t row = 1; row < JTK_NUM_ROW; row++) {
            const double v = total[p * JTK_NUM_ROW + row];
            if (v > g) {
                g = v;
                best = row;
            }
        }
        s_best[p] = (unsigned char)(best | (g > JTK_POLISH_MIN_GAIN ? 0x80u : 0u));
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const uint32_t round = st->rounds;
    const uint32_t inactive = 5 + (5 * round) % 21;
    Edit *edits = edits_all + cm.edit_off;
    uint32_t ne = 0, pos = ignore_edge, grow = 0;
    while (pos + ignore_edge < L) {
        const unsigned char b = s_best[pos];
        if ((b & 0x80u) && ne < cm.edit_cap) {
            const uint32_t row = b & 0x7fu;
            edits[ne].pos = pos;
            edits[ne].row = row;
            ne++;
            grow += edit_inserted(row, pos, L);
            pos += (row >= 11 ? row - 10 : 1) + inactive;
        } else {
            pos++;
        }
    }
    st->rounds = round + 1;
    st->n_edits = ne;
    if (ne == 0) {
        st->active = 0;
        return;
    }
    if (L + grow > cm.tmpl_cap) {
        st->status = JTK_ERR_CHUNK_FAILED;
        return;
    }
    // edited template into the other buffer
    const uint8_t *src = bufs.tmpl[st->buf] + cm.tmpl_off;
    uint8_t *dst = bufs.tmpl[JTK_NEXT_BUF(st->buf)] + cm.tmpl_off;
    uint32_t w = 0, e = 0, p = 0;
    while (p < L) {
        if (e < ne && edits[e].pos == p) {
            const uint32_t row = edits[e].row;
            e++;
            if (row < 4) {
                dst[w++] = (uint8_t)row;
                p++;
            } else if (row < 8) {
                dst[w++] = (uint8_t)(row - 4);
                dst[w++] = src[p++];
            } else if (row < 11) {
                const uint32_t c = row - 7;
                for (uint32_t q = 0; q < c && p + q < L; q++) dst[w++] = src[p + q];
                dst[w++] = src[p++];
            } else {
                p += row - 10;
            }
        } else {
            dst[w++] = src[p++];
        }
    }
    new_len[ci] = w;
}

// Local surgery of a read's ops around the round's edits, then Match / Mismatch re-tagging against the edited template.
// What the serial walk does (rounds 1-5: one thread per read, ~2,300 dependent steps, 2.3 ms per round however few reads are
// active), op by op with `ti` old template bases consumed:
//   * an Ins op passes through;
//   * an op that consumes old base ti inside a DELETION edit (pos <= ti < pos + d) becomes an Ins if it aligned a read base
//     (Match / Mismatch) and disappears if it was a Del; any other consuming op passes through;
//   * after the op that makes ti == pos of an INSERTION edit (rows 4..10; also before the first op for pos == 0), k Del ops
//     follow, k = edit_inserted();
//   * substitutions change no op; every Match / Mismatch is re-tagged from the new template base and the read base it aligns.
// Edits of a round are at least six positions apart (select_edits_kernel), so an op meets at most one deletion and one insertion
// edit, and every rule above is a function of the op and of ti alone -- ti is a prefix count.  Round 6: one WAVE per read, 64 ops
// per step: ballot / mbcnt for ti, one packed prefix sum for the output position and the new (template, read) coordinates of
// every emitted op.  Byte for byte the serial result (every full-path parity test compares the ops).
#define RT_WAVES 4
__global__ __launch_bounds__(64 * RT_WAVES) void rethread_kernel(uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                                                                ChunkState *state, DevBufs bufs, const uint8_t *ey_all,
                                                                const Edit *edits_all) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t r = blockIdx.x * RT_WAVES + (threadIdx.x >> 6);
    if (r >= n_reads) return;
    const ReadMeta rm = reads[r];
    ChunkState *st = &state[rm.chunk];
    if (st->status != 0 || !st->active || st->n_edits == 0) return;
    const ChunkMeta cm = chunks[rm.chunk];
    const uint32_t L = st->tmpl_len, ne = st->n_edits, b = st->buf, nb = JTK_NEXT_BUF(b);
    const Edit *edits = edits_all + cm.edit_off;
    const uint8_t *ops = bufs.ops[b] + rm.ops_off;
    uint8_t *out = bufs.ops[nb] + rm.ops_off;
    const uint32_t n_ops = bufs.ops_len[b][r];
    const uint8_t *tmpl = bufs.tmpl[nb] + cm.tmpl_off;  // the edited template
    const uint8_t *ey = ey_all + rm.ey_off;               // read base j is ey[j+1] & 3
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t w0 = 0, ti0 = 0, ni0 = 0, nj0 = 0;  // uniform: ops written, old template bases consumed, new template / read position
    uint32_t e_cur = 0;                          // uniform: edits before it cannot touch an op from here on
    bool overflow = false;
    // an insertion edit at position 0 fires before the first op
    if (edits[0].pos == 0 && edits[0].row >= 4 && edits[0].row < 11) {
        const uint32_t k = edit_inserted(edits[0].row, 0, L);
        if (lane < k && lane < rm.ops_cap) out[lane] = JTK_OP_DEL;
        overflow = overflow || k > rm.ops_cap;
        w0 = k;
        ni0 = k;
    }
    for (uint32_t base = 0; base < n_ops; base += 64) {
        const uint32_t kk = base + lane;
        const bool has = kk < n_ops;
        const uint32_t op = has ? ops[kk] : (uint32_t)JTK_OP_INS;
        const bool cons = has && op != JTK_OP_INS;                   // consumes an old template base
        const unsigned long long mc = __ballot(cons);
        const uint32_t ti = ti0 + (uint32_t)__popcll(mc & below);    // old bases consumed before this op
        // edits whose reach ends before this step's first consuming op are behind us (uniform cursor)
        while (e_cur < ne) {
            const Edit ed = edits[e_cur];
            const uint32_t last = ed.row >= 11 ? ed.pos + (ed.row - 10) - 1 : (ed.row >= 4 ? ed.pos - 1 : ed.pos);  // last ti it touches
            if ((ed.row >= 4 && ed.row < 11 && ed.pos == 0) || last < ti0)
                e_cur++;
            else
                break;
        }
        uint32_t first = has ? op : 4u;   // what the op becomes: an op code, or 4 = nothing
        uint32_t k_del = 0;               // Del ops that follow it
        if (cons) {
            for (uint32_t e = e_cur; e < ne; e++) {
                const Edit ed = edits[e];
                if (ed.pos > ti + 1) break;
                if (ed.row >= 11) {
                    if (ed.pos <= ti && ti < ed.pos + (ed.row - 10)) first = op != JTK_OP_DEL ? (uint32_t)JTK_OP_INS : 4u;
                } else if (ed.row >= 4 && ed.pos == ti + 1) {
                    k_del = edit_inserted(ed.row, ed.pos, L);
                }
            }
        }
        const uint32_t c_out = (first != 4u ? 1u : 0u) + k_del;
        const uint32_t c_t = ((first <= JTK_OP_MISMATCH || first == JTK_OP_DEL) ? 1u : 0u) + k_del;     // new template bases consumed
        const uint32_t c_r = (first <= JTK_OP_MISMATCH || first == JTK_OP_INS) ? 1u : 0u;                // read bases consumed
        const uint32_t packed = c_out | c_t << 10 | c_r << 20;
        uint32_t incl = packed;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += u;
        }
        const uint32_t excl = incl - packed, total = __shfl(incl, 63, 64);
        uint32_t w = w0 + (excl & 1023u);
        const uint32_t ni = ni0 + ((excl >> 10) & 1023u), nj = nj0 + (excl >> 20);
        if (first != 4u) {
            uint32_t o1 = first;
            if (first <= JTK_OP_MISMATCH) o1 = tmpl[ni] == (ey[nj + 1] & 3) ? JTK_OP_MATCH : JTK_OP_MISMATCH;
            if (w < rm.ops_cap) out[w] = (uint8_t)o1;
            w++;
        }
        for (uint32_t q = 0; q < k_del; q++)
            if (w + q < rm.ops_cap) out[w + q] = JTK_OP_DEL;
        w0 += total & 1023u;
        ni0 += (total >> 10) & 1023u;
        nj0 += total >> 20;
        ti0 += (uint32_t)__popcll(mc);
        overflow = overflow || w0 > rm.ops_cap;
    }
    if (lane == 0) {
        bufs.ops_len[nb][r] = overflow ? rm.ops_cap : w0;
        if (overflow) atomicMin(&st->status, (int)JTK_ERR_CHUNK_FAILED);
    }
}

// reads of chunks WITHOUT edits keep their ops: copy them so both buffers stay valid is unnecessary --
// only chunks with edits flip.
// One workgroup walks all chunks (a few words of state each), so the round's "chunks still active" count is known to one
// thread when the kernel ends: it goes straight to the host through mapped pinned memory -- no read-back blit, whose
// 4-byte copy kernel would wait for a wave slot behind the resident waves of the other streams.
__global__ __launch_bounds__(256) void commit_kernel(uint32_t n_chunks, ChunkState *state, const uint32_t *new_len,
                                                     uint32_t *n_active, uint32_t *n_active_host) {
    uint32_t mine = 0;
    for (uint32_t ci = threadIdx.x; ci < n_chunks; ci += 256) {
        ChunkState *st = &state[ci];
        if (st->status != 0 || !st->active) continue;
        if (st->n_edits > 0) {
            st->buf = JTK_NEXT_BUF(st->buf);
            st->tmpl_len = new_len[ci];
            st->n_edits = 0;
        }
        mine++;
    }
    __shared__ uint32_t s_total;
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    if (mine) atomicAdd(&s_total, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        *n_active = s_total;
        if (n_active_host) __hip_atomic_store(n_active_host, s_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// A pass starts from the batch as uploaded: every chunk's state from its pristine copy (buffer set 0, round 0, active) and
// the per-round "chunks still active" counters at zero.  One small kernel instead of the copy / fill blits a host-side
// reset costs (they queue behind the resident waves of other streams).
__global__ void reset_pass_kernel(uint32_t n_chunks, ChunkState *state, const ChunkState *state0, uint32_t *n_active,
                                  uint32_t n_counters) {
    const uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci < n_counters) n_active[ci] = 0;
    if (ci < n_chunks) state[ci] = state0[ci];
}

}  // namespace

void launch_reset_pass(hipStream_t s, uint32_t n_chunks, ChunkState *state, const ChunkState *state0, uint32_t *n_active,
                       uint32_t n_counters) {
    const uint32_t n = n_chunks > n_counters ? n_chunks : n_counters;
    reset_pass_kernel<<<(n + 127) / 128, 128, 0, s>>>(n_chunks, state, state0, n_active, n_counters);
}

void launch_polish_round(hipStream_t s, uint32_t n_chunks, uint32_t n_reads, const ReadMeta *reads,
                         const ChunkMeta *chunks, ChunkState *state, DevBufs bufs, const uint8_t *ey,
                         const double *table, double *total, Edit *edits, uint32_t *new_len, uint32_t max_tmpl,
                         uint32_t ignore_edge, int final_pass, uint32_t *n_active_out, uint32_t *n_active_host) {
    if (n_chunks == 0) return;
    // n_active_out is this round's own counter (commit_kernel stores it; no fill blit here); n_active_host, if given, is a
    // device-visible pointer into pinned host memory that receives the same number
    // (`total` holds the round's column totals: launch_sum_final, phmm_kernels.hip)
    (void)table;
    select_edits_kernel<<<n_chunks, 64, max_tmpl + 64, s>>>(chunks, state, bufs, total, edits, new_len,
                                                            ignore_edge, final_pass);
    if (!final_pass) {
        rethread_kernel<<<(n_reads + RT_WAVES - 1) / RT_WAVES, 64 * RT_WAVES, 0, s>>>(n_reads, reads, chunks, state, bufs, ey, edits);
        commit_kernel<<<1, 256, 0, s>>>(n_chunks, state, new_len, n_active_out, n_active_host);
    }
}
