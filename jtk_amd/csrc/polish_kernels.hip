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

// one thread per read: local surgery of the ops around the edits, then Match/Mismatch re-tagging.
__global__ void rethread_kernel(uint32_t n_reads, const ReadMeta *reads, const ChunkMeta *chunks,
                                ChunkState *state, DevBufs bufs, const uint8_t *ey_all, const Edit *edits_all) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
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
    const uint8_t *ey = ey_all + rm.ey_off;                // read base j is ey[j+1] & 3
    uint32_t w = 0, e = 0, ti = 0;
    uint32_t ni = 0, nj = 0;  // positions in the NEW template / the read, for re-tagging
    bool overflow = false;
    // One thread per read, latency bound: the three input streams (ops, new template, read) are fetched 8 bytes
    // at a time and the output leaves 8 bytes at a time (slots are 8-byte aligned with >= 8 bytes of slack).
    uint64_t in8 = 0, t8 = 0, y8 = 0, out8 = 0;
    uint32_t t8_at = ~0u, y8_at = ~0u;  // 8-byte block currently held of tmpl / ey
    auto tmpl_at = [&](uint32_t q) -> uint8_t {
        const uint64_t a = (uint64_t)(cm.tmpl_off + q);  // absolute byte offset: blocks are aligned in the buffer
        const uint32_t blk = (uint32_t)(a >> 3);
        if (blk != t8_at) {
            t8 = *reinterpret_cast<const uint64_t *>(bufs.tmpl[nb] + ((uint64_t)blk << 3));
            t8_at = blk;
        }
        return (uint8_t)(t8 >> (8 * (a & 7)));
    };
    auto ey_at = [&](uint32_t q) -> uint8_t {
        const uint64_t a = rm.ey_off + q;
        const uint32_t blk = (uint32_t)(a >> 3);
        if (blk != y8_at) {
            y8 = *reinterpret_cast<const uint64_t *>(ey_all + ((uint64_t)blk << 3));
            y8_at = blk;
        }
        return (uint8_t)(y8 >> (8 * (a & 7)));
    };
    auto emit = [&](uint8_t op) {
        if (w >= rm.ops_cap) {
            overflow = true;
            return;
        }
        if (op == JTK_OP_INS) {
            nj++;
        } else if (op == JTK_OP_DEL) {
            ni++;
        } else {
            op = tmpl_at(ni) == (ey_at(nj + 1) & 3) ? JTK_OP_MATCH : JTK_OP_MISMATCH;
            ni++;
            nj++;
        }
        out8 |= (uint64_t)op << (8 * (w & 7u));
        w++;
        if ((w & 7u) == 0) {
            *reinterpret_cast<uint64_t *>(out + w - 8) = out8;
            out8 = 0;
        }
    };
    auto pending_inserts = [&]() {
        while (e < ne && edits[e].pos == ti && edits[e].row >= 4 && edits[e].row < 11) {
            const uint32_t k = edit_inserted(edits[e].row, edits[e].pos, L);
            for (uint32_t q = 0; q < k; q++) emit(JTK_OP_DEL);
            e++;
        }
    };
    pending_inserts();
    for (uint32_t k = 0; k < n_ops; k++) {
        if ((k & 7u) == 0) in8 = *reinterpret_cast<const uint64_t *>(ops + k);
        const uint8_t op = (uint8_t)(in8 >> (8 * (k & 7u)));
        if (op == JTK_OP_INS) {
            emit(op);
            continue;
        }
        if (e < ne && edits[e].row >= 11 && edits[e].pos <= ti && ti < edits[e].pos + (edits[e].row - 10)) {
            if (op != JTK_OP_DEL) emit(JTK_OP_INS);
            ti++;
            if (ti == edits[e].pos + (edits[e].row - 10)) e++;
        } else {
            emit(op);
            if (e < ne && edits[e].row < 4 && edits[e].pos == ti) e++;
            ti++;
        }
        pending_inserts();
    }
    if ((w & 7u) != 0 && !overflow) *reinterpret_cast<uint64_t *>(out + (w & ~7u)) = out8;  // inside the slot: cap % 8 == 0
    bufs.ops_len[nb][r] = w;
    if (overflow) atomicMin(&st->status, (int)JTK_ERR_CHUNK_FAILED);
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
        rethread_kernel<<<(n_reads + 63) / 64, 64, 0, s>>>(n_reads, reads, chunks, state, bufs, ey, edits);
        commit_kernel<<<1, 256, 0, s>>>(n_chunks, state, new_len, n_active_out, n_active_host);
    }
}
