// host_api.cpp -- the host-only entry points of include/jtk_lc.h (no GPU needed).
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jtk_lc.h"

extern "C" {

const char *jtk_lc_strerror(int status) {
    switch (status) {
        case JTK_OK: return "ok";
        case JTK_ERR_INVALID_ARG: return "invalid argument";
        case JTK_ERR_NO_DEVICE: return "no usable gfx950 device / HIP runtime error";
        case JTK_ERR_UNSUPPORTED: return "unsupported configuration (band radius > 127, or pile-up too large)";
        case JTK_ERR_ALLOC: return "allocation failed";
        case JTK_ERR_OPS_MISMATCH: return "alignment ops do not consume exactly the template and the read";
        case JTK_ERR_CHUNK_FAILED: return "a chunk failed where the reference would panic";
        case JTK_ERR_INTERNAL: return "internal error";
        default: return "unknown status";
    }
}

int jtk_lc_version(void) { return JTK_LC_ABI_VERSION; }

// Sort key of pileup_nodes (local_clustering/mod.rs:47-50): the number of alignment columns of
// Node::recover (definitions/src/lib.rs:773-813) whose symbol is not '|': every Ins/Del column plus
// every aligned column whose bases differ case-insensitively.  The reference cigar has no mismatch op
// (definitions/src/lib.rs:816-823), so aligned columns compare bases, not the Match/Mismatch tag.
int jtk_lc_pileup_sort_key(const uint8_t *tmpl, uint64_t tmpl_len, const uint8_t *read, uint64_t read_len,
                           const uint8_t *ops, uint64_t ops_len, uint64_t *key_out) {
    if (!key_out || (ops_len && !ops)) return JTK_ERR_INVALID_ARG;
    uint64_t q = 0, r = 0, key = 0;
    for (uint64_t i = 0; i < ops_len; i++) {
        switch (ops[i]) {
            case JTK_OP_DEL:
                key++;
                r++;
                break;
            case JTK_OP_INS:
                key++;
                q++;
                break;
            case JTK_OP_MATCH:
            case JTK_OP_MISMATCH:
                if (q >= read_len || r >= tmpl_len) return JTK_ERR_OPS_MISMATCH;
                if ((read[q] & 0xdf) != (tmpl[r] & 0xdf)) key++;
                q++;
                r++;
                break;
            default:
                return JTK_ERR_INVALID_ARG;
        }
    }
    if (q != read_len || r != tmpl_len) return JTK_ERR_OPS_MISMATCH;
    *key_out = key;
    return JTK_OK;
}

// normalize_local_clustering for one pile-up (local_clustering/normalize.rs:26-49): clusters are
// renumbered by descending size; `sort_by_key` is stable and the list is then reversed, so among equal
// sizes the LARGER old index gets the smaller new label.  Each posterior row is permuted the same way
// (reorder, normalize.rs:54-63).
int jtk_lc_normalize_pileup(uint32_t n_reads, uint32_t cluster_num, uint32_t *label, double *log_post,
                            uint32_t post_stride) {
    if (cluster_num == 0) return JTK_OK;
    if (!label || !log_post || cluster_num > post_stride) return JTK_ERR_INVALID_ARG;
    std::vector<uint32_t> count(cluster_num, 0), order(cluster_num), mapsto(cluster_num);
    for (uint32_t i = 0; i < n_reads; i++) {
        if (label[i] >= cluster_num) return JTK_ERR_INVALID_ARG;
        count[label[i]]++;
    }
    for (uint32_t c = 0; c < cluster_num; c++) order[c] = c;
    for (uint32_t a = 1; a < cluster_num; a++) {  // stable insertion sort, ascending by count
        const uint32_t f = order[a];
        uint32_t b = a;
        while (b > 0 && count[order[b - 1]] > count[f]) {
            order[b] = order[b - 1];
            b--;
        }
        order[b] = f;
    }
    for (uint32_t to = 0; to < cluster_num; to++) mapsto[order[cluster_num - 1 - to]] = to;
    std::vector<double> tmp(cluster_num);
    for (uint32_t i = 0; i < n_reads; i++) {
        double *row = log_post + (size_t)i * post_stride;
        for (uint32_t c = 0; c < cluster_num; c++) tmp[mapsto[c]] = row[c];
        memcpy(row, tmp.data(), cluster_num * sizeof(double));
        label[i] = mapsto[label[i]];
    }
    return JTK_OK;
}

}  // extern "C"
