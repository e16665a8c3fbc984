// io_kernels.hip -- the two ends of a stage call, on the device (round 6).
//
// The drop-in call (jtk_lc_cluster_chunks: the loop of haplotyper/src/local_clustering/mod.rs:63-81 as one batch) hands over
// host buffers as the Rust side holds them: ASCII bases, one op per byte, reads back to back.  Until round 5 the host recoded
// them into freshly allocated vectors before the upload (validation, `y | ctx << 2`, op slots) and unpacked the results base
// by base after a download of up to three whole buffer sets; 0.6 s of a 1.48 s call.  Now the raw bytes cross the bus once,
// straight from / into the caller's memory, and both conversions are kernels:
//   * encode_reads_kernel: raw read bases + raw ops -> d_ey (emission index of every read column), the ops slots of buffer
//     set 0 (zero padded to their capacity) and a validation word (bit 1: non-ACGT base in a read, bit 2: op code > 3);
//   * out_len_kernel / gather_kernel: a chunk's final consensus (2-bit codes -> ASCII) and the re-threaded ops of its reads,
//     from whichever buffer set the chunk's state names, packed back to back at offsets the host prefix-summed from the lengths.
// HBM-bound byte work (one pass over ~4 KB per read either way); trivial beside the sweeps.
#include "device_common.h"

namespace {

__device__ __forceinline__ int base_code_dev(uint32_t c) {  // session.hip base_code: ACGT in either case, else -1
    c &= 0xdfu;                                              // upper case (only letters can map onto letters this way)
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}

// one workgroup per read
__global__ __launch_bounds__(256) void encode_reads_kernel(uint32_t n_reads, const ReadMeta *reads, const uint8_t *raw_bases,
                                                           const uint64_t *raw_base_off, const uint8_t *raw_ops,
                                                           const uint64_t *raw_ops_off, uint8_t *ey_all, uint8_t *ops_all,
                                                           uint32_t *flags) {
    const uint32_t g = blockIdx.x;
    if (g >= n_reads) return;
    const ReadMeta rm = reads[g];
    const uint8_t *rb = raw_bases + (raw_base_off[g] - raw_base_off[0]);
    uint8_t *ey = ey_all + rm.ey_off;
    uint32_t bad = 0;
    for (uint32_t j = threadIdx.x; j <= rm.read_len; j += blockDim.x) {
        if (j == 0) {
            ey[0] = 0;  // (unused: column 0 emits nothing)
            continue;
        }
        const int code = base_code_dev(rb[j - 1]);
        const int prev = j >= 2 ? base_code_dev(rb[j - 2]) & 3 : 4;
        if (code < 0) bad |= 2u;
        ey[j] = (uint8_t)((code & 3) | (prev << 2));
    }
    const uint8_t *src = raw_ops + (raw_ops_off[g] - raw_ops_off[0]);
    const uint32_t len = (uint32_t)(raw_ops_off[g + 1] - raw_ops_off[g]);
    uint8_t *dst = ops_all + rm.ops_off;
    for (uint32_t k = threadIdx.x; k < rm.ops_cap; k += blockDim.x) {
        const uint8_t v = k < len ? src[k] : (uint8_t)0;
        if (v > JTK_OP_DEL) bad |= 4u;
        dst[k] = v;
    }
    if (bad) atomicOr(flags, bad);
}

// per read: the length of its re-threaded ops (0 for a failed chunk); per chunk: the consensus length
__global__ void out_len_kernel(uint32_t n_reads, uint32_t n_chunks, const ReadMeta *reads, const ChunkState *state, DevBufs bufs,
                               uint32_t *ops_len_out, uint32_t *cons_len_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_reads) {
        const ChunkState cs = state[reads[i].chunk];
        ops_len_out[i] = cs.status == 0 ? bufs.ops_len[cs.buf % 3u][i] : 0u;
    }
    if (i < n_chunks) {
        const ChunkState cs = state[i];
        cons_len_out[i] = cs.status == 0 ? cs.tmpl_len : 0u;
    }
}

// blocks [0, n_reads): one read's ops; blocks [n_reads, n_reads + n_chunks): one chunk's consensus as ASCII
__global__ __launch_bounds__(256) void gather_kernel(uint32_t n_reads, uint32_t n_chunks, const ReadMeta *reads,
                                                     const ChunkMeta *chunks, const ChunkState *state, DevBufs bufs,
                                                     const uint64_t *ops_out_off, const uint64_t *cons_off, uint8_t *ops_out,
                                                     uint8_t *cons_out) {
    const uint32_t b = blockIdx.x;
    if (b < n_reads) {
        if (!ops_out) return;
        const ReadMeta rm = reads[b];
        const ChunkState cs = state[rm.chunk];
        if (cs.status != 0) return;
        const uint32_t len = (uint32_t)(ops_out_off[b + 1] - ops_out_off[b]);
        const uint8_t *src = bufs.ops[cs.buf % 3u] + rm.ops_off;
        uint8_t *dst = ops_out + ops_out_off[b];
        for (uint32_t k = threadIdx.x; k < len; k += blockDim.x) dst[k] = src[k];
        return;
    }
    const uint32_t c = b - n_reads;
    if (c >= n_chunks || !cons_out) return;
    const ChunkState cs = state[c];
    if (cs.status != 0) return;
    const uint8_t *src = bufs.tmpl[cs.buf % 3u] + chunks[c].tmpl_off;
    uint8_t *dst = cons_out + cons_off[c];
    for (uint32_t p = threadIdx.x; p < cs.tmpl_len; p += blockDim.x) dst[p] = (uint8_t)("ACGT"[src[p] & 3u]);
}

}  // namespace

void launch_encode_reads(hipStream_t s, uint32_t n_reads, const ReadMeta *reads, const uint8_t *raw_bases,
                         const uint64_t *raw_base_off, const uint8_t *raw_ops, const uint64_t *raw_ops_off, uint8_t *ey,
                         uint8_t *ops, uint32_t *flags) {
    if (n_reads == 0) return;
    encode_reads_kernel<<<n_reads, 256, 0, s>>>(n_reads, reads, raw_bases, raw_base_off, raw_ops, raw_ops_off, ey, ops, flags);
}

void launch_out_len(hipStream_t s, uint32_t n_reads, uint32_t n_chunks, const ReadMeta *reads, const ChunkState *state,
                    DevBufs bufs, uint32_t *ops_len_out, uint32_t *cons_len_out) {
    const uint32_t n = n_reads > n_chunks ? n_reads : n_chunks;
    if (n == 0) return;
    out_len_kernel<<<(n + 255) / 256, 256, 0, s>>>(n_reads, n_chunks, reads, state, bufs, ops_len_out, cons_len_out);
}

void launch_gather(hipStream_t s, uint32_t n_reads, uint32_t n_chunks, const ReadMeta *reads, const ChunkMeta *chunks,
                   const ChunkState *state, DevBufs bufs, const uint64_t *ops_out_off, const uint64_t *cons_off,
                   uint8_t *ops_out, uint8_t *cons_out) {
    if (n_reads + n_chunks == 0) return;
    gather_kernel<<<n_reads + n_chunks, 256, 0, s>>>(n_reads, n_chunks, reads, chunks, state, bufs, ops_out_off, cons_off, ops_out,
                                                     cons_out);
}
