// correction.hip -- jtk_lc_correct_clustering: the first consumer of the stage's posteriors.
//
// Replaces `AlignmentCorrection::correct_clustering_selected` (haplotyper/src/phmm_likelihood_correction.rs:32-97).
// Per selected chunk with more than one cluster:
//   similarity fill (:272-285)  N x N `alignment` (:466-479): two affine-gap alignments of neighbouring nodes' posteriors
//                               (`align_swg` :482-531 with `sim` :534-550 as the match score) + the centre's `sim`
//                               -> ONE DEVICE KERNEL over every ordered pair of every chunk (the O(N^2 ctx^2) part);
//   spectral clustering         filter_similarity :330-347, graph Laplacian :385-402, eigenvectors with eigenvalue below 0.2
//                               :405-464 (nalgebra in the reference; include/jtk_eigen.h here), posteriors appended, columns
//                               normalised, 20 x misc::kmeans on Xoroshiro128PlusPlus(id * k) -> host threads, one chunk each;
//   decision                    adjusted Rand index against the previous labels on biased reads :220-240, suppression of the
//                               lowest 5 % :100-105 unless the chunk is protected by its local-clustering score :108-129.
// Sums, sorts and tie-breaks follow the reference (stable sorts, `max_by` = last maximum, `min_by` = first minimum); the CPU
// oracle (oracle/correction.c) is the line-by-line restatement this file is tested against.  Where the reference panics the
// call returns JTK_ERR_CHUNK_FAILED and writes nothing.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "device_common.h"
#include "jtk_lc_debug.h"
#include "jtk_eigen.h"

extern "C" void jtk_internal_set_error(const char *msg);

namespace {

struct ArmEnt {  // one neighbouring node of a context
    uint32_t chunk;     // dense chunk index
    uint32_t post_len;
    uint64_t post_off;
};
struct Member {  // one (idx, read) of correct_chunk's `reads`, as the kernel sees it
    uint64_t up_off, down_off;  // into the arm entries
    uint32_t n_up, n_down;
    uint64_t post_off;          // the centre's posterior
    uint32_t post_len;
    uint32_t pad;
};
struct PairJob {  // one chunk's block of the similarity fill
    uint64_t first_pair;  // index of its first ordered pair in the global enumeration
    uint64_t sims_off;    // into the output
    uint32_t member0, n;
    uint32_t chunk;       // dense chunk index of the centre
    uint32_t pad;
};

__device__ __forceinline__ double d_max3(double a, double b, double c) { return jtk_fmax(jtk_fmax(a, b), c); }

// logit_from_lnp :553-566 (ln_1p as jtk_log(1 + x)); *panic is set where the reference asserts
__device__ double d_logit_from_lnp(double lnp, int *panic) {
    if (!(lnp <= 0.0)) *panic = 1;
    if (lnp < -80.0) return -80.0;
    if (-1.8e-35 < lnp) return 80.0;
    return lnp - jtk_log(1.0 + (-jtk_exp(lnp)));
}
// sim :534-550 with the streaming LogSumExp of misc.rs:94-140
__device__ double d_sim(const double *xs, uint32_t nx, const double *ys, uint32_t ny, const double *cps, uint32_t nc, int *panic) {
    if (nx != nc || nx != ny) {
        *panic = 1;
        return 0.0;
    }
    if (nc == 1) {
        double total = 0.0;
        total += cps[0];
        return -jtk_log(jtk_fmax(total, 1.5) - 1.0);
    }
    double accum = 0.0, mx = -__builtin_inf();
    for (uint32_t i = 0; i < nc; i++) {
        const double rhs = xs[i] + ys[i] - jtk_log(cps[i]);
        if (rhs < mx) {
            accum = accum + jtk_exp(rhs - mx);
        } else {
            accum = accum * jtk_exp(mx - rhs) + 1.0;
            mx = rhs;
        }
    }
    const double logp = jtk_log(accum) + mx;
    const double logit = d_logit_from_lnp(logp, panic);
    if (logit == __builtin_inf() || logit == -__builtin_inf()) *panic = 1;
    return logit;
}
// align_swg :482-531 with two rolling rows of (len2 + 1) x 3 doubles in `rows`
__device__ double d_align_swg(const ArmEnt *arm1, uint32_t len1, const ArmEnt *arm2, uint32_t len2, const double *post,
                              const double *cn, const uint64_t *cn_off, const uint32_t *cn_len, double *rows, int *panic) {
    const double GAP_OPEN = -0.5, GAP_EXTEND = -100.0, MISM = -100.0;
    const double lower = (double)(len1 + len2 + 2) * MISM;
    const uint32_t W = len2 + 1;
    double *prev = rows, *cur = rows + (size_t)3 * W;
    // row 0
    for (uint32_t j = 0; j <= len2; j++) {
        prev[3 * j] = lower;
        prev[3 * j + 1] = j >= 1 ? GAP_OPEN + (double)(j - 1) * GAP_EXTEND : lower;
        prev[3 * j + 2] = lower;
    }
    prev[0] = 0.0;
    // the last column (j == len2) of every row and the whole last row feed the result; `best` follows the reference's
    // chain: first the last ROW left to right, then the last COLUMN top to bottom (last maximum) -- so the column
    // candidates are kept until the last row is known
    double colbest = 0.0;  // running `max_by` over the last column's rows 0 .. i (applied after the row part below)
    // The chain is row_last (needs row len1) THEN column_last (rows 0..len1): evaluate the column part into a small
    // running state that can be replayed after the row part: max_by keeps the LAST maximum, so the final answer is
    //   m = max over all candidates; among equal maxima the last in chain order -- the value is the same either way.
    // Only the VALUE is returned, so the order of equal maxima is immaterial: a plain maximum over both sets.
    bool have = false;
    {
        const double x = d_max3(prev[3 * len2], prev[3 * len2 + 1], prev[3 * len2 + 2]);
        colbest = x;
        have = true;
    }
    for (uint32_t i = 1; i <= len1; i++) {
        cur[0] = lower;
        cur[1] = lower;
        cur[2] = GAP_OPEN + (double)(i - 1) * GAP_EXTEND;
        const ArmEnt a = arm1[i - 1];
        for (uint32_t j = 1; j <= len2; j++) {
            const ArmEnt b = arm2[j - 1];
            double match_score = MISM;
            if (a.chunk == b.chunk)
                match_score = d_sim(post + a.post_off, a.post_len, post + b.post_off, b.post_len, cn + cn_off[a.chunk],
                                    cn_len[a.chunk], panic);
            const double mat = d_max3(prev[3 * (j - 1)], prev[3 * (j - 1) + 1], prev[3 * (j - 1) + 2]) + match_score;
            const double del2 = jtk_fmax(jtk_fmax(cur[3 * (j - 1)] + GAP_OPEN, cur[3 * (j - 1) + 1] + GAP_EXTEND),
                                         cur[3 * (j - 1) + 2] + GAP_OPEN);
            const double del1 = jtk_fmax(jtk_fmax(prev[3 * j] + GAP_OPEN, prev[3 * j + 1] + GAP_OPEN), prev[3 * j + 2] + GAP_EXTEND);
            cur[3 * j] = mat;
            cur[3 * j + 1] = del2;
            cur[3 * j + 2] = del1;
        }
        const double x = d_max3(cur[3 * len2], cur[3 * len2 + 1], cur[3 * len2 + 2]);
        if (!have || !(x < colbest)) colbest = x, have = true;
        double *t = prev;
        prev = cur;
        cur = t;
    }
    double best = colbest;
    for (uint32_t j = 0; j <= len2; j++) {  // the last row (now in `prev`)
        const double x = d_max3(prev[3 * j], prev[3 * j + 1], prev[3 * j + 2]);
        if (!(x < best)) best = x;
    }
    return best;
}

// every ordered pair (i, j), i != j, of every job: sims[i][j] = alignment(ctx_i, ctx_j) (:466-479)
__global__ void similarity_kernel(uint64_t n_pairs, uint32_t n_jobs, const PairJob *jobs, const Member *members, const ArmEnt *arms,
                                  const double *post, const double *cn, const uint64_t *cn_off, const uint32_t *cn_len,
                                  double *sims, double *scratch, uint32_t row_doubles, int *panic_out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    double *rows = scratch + tid * (uint64_t)row_doubles;
    int panic = 0;
    uint32_t job = 0;
    for (uint64_t p = tid; p < n_pairs; p += stride) {
        while (job + 1 < n_jobs && jobs[job + 1].first_pair <= p) job++;
        while (job > 0 && jobs[job].first_pair > p) job--;
        const PairJob jb = jobs[job];
        const uint64_t q = p - jb.first_pair;
        const uint32_t i = (uint32_t)(q / jb.n), j = (uint32_t)(q % jb.n);
        double out = 0.0;
        if (i != j) {
            const Member a = members[jb.member0 + i], b = members[jb.member0 + j];
            const double up = d_align_swg(arms + a.up_off, a.n_up, arms + b.up_off, b.n_up, post, cn, cn_off, cn_len, rows, &panic);
            const double down =
                d_align_swg(arms + a.down_off, a.n_down, arms + b.down_off, b.n_down, post, cn, cn_off, cn_len, rows, &panic);
            const double center = d_sim(post + a.post_off, a.post_len, post + b.post_off, b.post_len, cn + cn_off[jb.chunk],
                                        cn_len[jb.chunk], &panic);
            const double lr = up + down + center;
            out = 1.0 / (1.0 + jtk_exp(-lr));
        }
        sims[jb.sims_off + (uint64_t)i * jb.n + j] = out;
    }
    if (panic) atomicOr(panic_out, 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// host: random sampling of rand 0.8.5 on rand_xoshiro 0.6.0's Xoroshiro128PlusPlus (phmm_likelihood_correction.rs:299-301)
// and misc::kmeans (misc.rs:229-341), as the reference calls them
// ---------------------------------------------------------------------------------------------------------------------
struct Rng128 {
    uint64_t s0, s1;
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    explicit Rng128(uint64_t seed) {  // seed_from_u64: two SplitMix64 outputs
        auto sm = [&]() {
            seed += 0x9e3779b97f4a7c15ULL;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
            return z ^ (z >> 31);
        };
        s0 = sm();
        s1 = sm();
    }
    uint64_t next_u64() {
        const uint64_t a = s0;
        uint64_t b = s1;
        const uint64_t r = rotl(a + b, 17) + a;
        b ^= a;
        s0 = rotl(a, 49) ^ b ^ (b << 21);
        s1 = rotl(b, 28);
        return r;
    }
    uint32_t next_u32() { return (uint32_t)next_u64(); }
    uint64_t gen_range_usize(uint64_t n) {
        const uint64_t zone = (n << __builtin_clzll(n)) - 1;
        for (;;) {
            const uint64_t v = next_u64();
            const unsigned __int128 m = (unsigned __int128)v * n;
            if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
        }
    }
    uint64_t gen_index(uint64_t ub) {
        if (ub > 0xffffffffULL) return gen_range_usize(ub);
        const uint32_t n = (uint32_t)ub, zone = (n << __builtin_clz(n)) - 1;
        for (;;) {
            const uint64_t m = (uint64_t)next_u32() * n;
            if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
        }
    }
    bool gen_bool(double p) {
        if (p == 1.0) return true;
        const double scaled = p * 18446744073709551616.0;
        const uint64_t p_int = !(scaled > 0.0) ? 0 : (scaled >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)scaled);
        return next_u64() < p_int;
    }
    // SliceRandom::choose_weighted -> WeightedIndex<f64>; -1 on WeightedError
    long choose_weighted(const std::vector<double> &w) {
        const size_t n = w.size();
        if (n == 0 || !(w[0] >= 0.0)) return -1;
        std::vector<double> cum(n > 1 ? n - 1 : 0);
        double total = w[0];
        for (size_t i = 1; i < n; i++) {
            if (!(w[i] >= 0.0)) return -1;
            cum[i - 1] = total;
            total += w[i];
        }
        if (total == 0.0) return -1;
        double scale = total;
        const double max_rand = 1.0 - 0x1p-52;
        while (scale * max_rand + 0.0 >= total) {
            uint64_t b;
            memcpy(&b, &scale, 8);
            b -= 1;
            memcpy(&scale, &b, 8);
        }
        const uint64_t bits = (next_u64() >> 12) | 0x3ff0000000000000ULL;
        double v12;
        memcpy(&v12, &bits, 8);
        const double chosen = (v12 - 1.0) * scale + 0.0;
        size_t lo = 0, hi = n - 1;
        while (lo < hi) {
            const size_t mid = lo + (hi - lo) / 2;
            if (cum[mid] <= chosen)
                lo = mid + 1;
            else
                hi = mid;
        }
        return (long)lo;
    }
};

double row_dist(const double *a, const double *b, size_t dim) {
    double s = 0.0;
    for (size_t d = 0; d < dim; d++) {
        const double t = a[d] - b[d];
        s += t * t;
    }
    return s;
}
// misc::kmeans (misc.rs:229-259): false where the reference panics
bool kmeans(const std::vector<double> &data, size_t n, size_t dim, size_t k, Rng128 &rng, double *dist_out, std::vector<size_t> &assign) {
    const double UPDATE_THR = 0.00000001;
    if (k < 1 || dim == 0) return false;
    assign.assign(n, 0);
    auto nearest = [&](const std::vector<const double *> &centers) {  // update_assignments :261-276 (first minimum)
        for (size_t i = 0; i < n; i++) {
            size_t best = 0;
            double bd = row_dist(&data[i * dim], centers[0], dim);
            for (size_t c = 1; c < centers.size(); c++) {
                const double d = row_dist(&data[i * dim], centers[c], dim);
                if (d < bd) {
                    bd = d;
                    best = c;
                }
            }
            assign[i] = best;
        }
    };
    if (rng.gen_bool(0.5)) {
        for (size_t i = 0; i < n; i++) assign[i] = (size_t)rng.gen_range_usize(k);
    } else {  // suggest_first :315-341
        if (k > n) return false;
        std::vector<const double *> centers;
        centers.push_back(&data[rng.gen_index(n) * dim]);
        std::vector<double> dists(n);
        for (size_t it = 0; it + 1 < k; it++) {
            for (size_t i = 0; i < n; i++) {
                double m = row_dist(&data[i * dim], centers[0], dim);
                for (size_t c = 1; c < centers.size(); c++) {
                    const double d = row_dist(&data[i * dim], centers[c], dim);
                    if (d < m) m = d;
                }
                dists[i] = m;
            }
            const long idx = rng.choose_weighted(dists);
            if (idx < 0) return false;
            centers.push_back(&data[(size_t)idx * dim]);
        }
        nearest(centers);
    }
    std::vector<double> centers(k * dim, 0.0);
    std::vector<size_t> counts(k);
    std::vector<const double *> cptr(k);
    for (size_t c = 0; c < k; c++) cptr[c] = &centers[c * dim];
    auto get_dist = [&]() {
        double s = 0.0;
        for (size_t i = 0; i < n; i++) s += row_dist(&data[i * dim], &centers[assign[i] * dim], dim);
        return s;
    };
    double d = get_dist();
    for (;;) {
        std::fill(centers.begin(), centers.end(), 0.0);  // update_centers :277-297
        std::fill(counts.begin(), counts.end(), 0);
        for (size_t i = 0; i < n; i++) {
            double *c = &centers[assign[i] * dim];
            for (size_t q = 0; q < dim; q++) c[q] += data[i * dim + q];
            counts[assign[i]]++;
        }
        for (size_t c = 0; c < k; c++)
            if (counts[c] > 0)
                for (size_t q = 0; q < dim; q++) centers[c * dim + q] /= (double)counts[c];
        nearest(cptr);
        const double nd = get_dist();
        if (!(nd < d + UPDATE_THR)) return false;  // assert!(new_dist < dist + UPDATE_THR)
        if (d - nd < UPDATE_THR) break;
        d = nd;
    }
    *dist_out = d;
    return true;
}

double logsumexp(const double *xs, size_t n) {  // misc.rs:84-92
    if (n == 0) return 0.0;
    double mx = xs[0];
    for (size_t i = 1; i < n; i++)
        if (!(xs[i] < mx)) mx = xs[i];
    double sum = 0.0;
    for (size_t i = 0; i < n; i++) sum += jtk_exp(xs[i] - mx);
    return mx + jtk_log(sum);
}
double round_half_away(double x) { return x < 0.0 ? -std::floor(-x + 0.5) : std::floor(x + 0.5); }

// misc.rs:22-46; *panic where the reference does
double adjusted_rand_index(const std::vector<size_t> &label, const std::vector<size_t> &pred, bool *panic) {
    const size_t n = label.size();
    if (n == 0) {
        *panic = true;
        return 0.0;
    }
    const size_t lab_max = *std::max_element(label.begin(), label.end()), pred_max = *std::max_element(pred.begin(), pred.end());
    std::vector<size_t> cont((lab_max + 1) * (pred_max + 1), 0), lab_sum(lab_max + 1, 0), pred_sum(pred_max + 1, 0);
    for (size_t i = 0; i < n; i++) {
        cont[label[i] * (pred_max + 1) + pred[i]]++;
        lab_sum[label[i]]++;
        pred_sum[pred[i]]++;
    }
    auto choose = [](size_t x) { return (std::max<size_t>(x, 1) - 1) * x / 2; };
    size_t lab_match = 0, pred_match = 0, both_match = 0;
    for (size_t x : lab_sum) lab_match += choose(x);
    for (size_t x : pred_sum) pred_match += choose(x);
    const size_t num_of_pairs = choose(n);
    for (size_t x : cont) both_match += choose(x);
    if (!(both_match <= (lab_match + pred_match) / 2)) *panic = true;
    const int64_t match_prod = (int64_t)(lab_match * pred_match);
    const int64_t denom = (int64_t)(num_of_pairs * (lab_match + pred_match) / 2) - match_prod;
    const int64_t numer = (int64_t)(num_of_pairs * both_match) - match_prod;
    return (double)numer / (double)denom;
}

int cc_fail(int status, const std::string &msg) {
    jtk_internal_set_error(msg.c_str());
    return status;
}
#define CC_HIP(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            return cc_fail(_e == hipErrorOutOfMemory ? JTK_ERR_ALLOC : JTK_ERR_NO_DEVICE,                    \
                           std::string(#expr) + ": " + hipGetErrorString(_e));                               \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    template <typename T>
    int upload(const std::vector<T> &v, hipStream_t st) {
        if (hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)) != hipSuccess) return JTK_ERR_ALLOC;
        if (!v.empty() && hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st) != hipSuccess)
            return JTK_ERR_NO_DEVICE;
        return 0;
    }
};

thread_local bool g_keep_sims = false;
thread_local std::vector<double> g_first_sims;

}  // namespace

// test hooks (include/jtk_lc_debug.h, not the boundary header): keep the raw similarity matrix of the first corrected chunk of the next calls on this
// thread, and read it back -- the device fill is compared bit for bit with the oracle's
extern "C" void jtk_lc_debug_cc_keep_sims(int on) { g_keep_sims = on != 0; }
extern "C" size_t jtk_lc_debug_cc_first_sims(double *out, size_t cap) {
    const size_t n = std::min(cap, g_first_sims.size());
    if (out && n) memcpy(out, g_first_sims.data(), n * sizeof(double));
    return g_first_sims.size();
}

extern "C" int jtk_lc_correct_clustering(size_t n_reads, const uint64_t *read_id, const uint64_t *node_off,
                                         const jtk_cc_node_t *nodes, const double *posteriors, size_t n_chunks,
                                         jtk_cc_chunk_t *chunks, size_t n_selected, const uint64_t *selection,
                                         double haploid_coverage, double min_gain, uint64_t *cluster_out, uint8_t *touched,
                                         int device) {
    jtk_internal_set_error("");
    (void)read_id;  // the reference keys its write-back by read id; positions in the flattened arrays are the same thing
    if (!node_off || (n_reads && !nodes) || !chunks || !cluster_out || !touched || (n_selected && !selection))
        return cc_fail(JTK_ERR_INVALID_ARG, "null argument");
    const size_t n_nodes = (size_t)node_off[n_reads];
    for (size_t e = 0; e < n_nodes; e++) {
        cluster_out[e] = nodes[e].cluster;
        touched[e] = 0;
    }
    if (n_chunks == 0) return cc_fail(JTK_ERR_CHUNK_FAILED, "no chunk (the reference unwraps the largest chunk id)");
    {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
            return cc_fail(JTK_ERR_NO_DEVICE, "no usable HIP device (jtk_lc has no CPU fallback)");
        CC_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        CC_HIP(hipGetDeviceProperties(&prop, device));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            return cc_fail(JTK_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    }
    bool panic = false;
    // dense chunk index; the reference indexes plain vectors by chunk id (:131-137)
    std::unordered_map<uint64_t, uint32_t> dense;
    uint64_t max_id = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        if (!dense.emplace(chunks[c].id, (uint32_t)c).second) return cc_fail(JTK_ERR_INVALID_ARG, "chunk ids repeat");
        max_id = std::max(max_id, chunks[c].id);
    }
    std::vector<uint32_t> node_chunk(n_nodes);
    std::vector<uint32_t> coverage(n_chunks, 0);
    for (size_t e = 0; e < n_nodes; e++) {
        auto it = dense.find(nodes[e].chunk);
        if (it == dense.end()) {
            // a node on a chunk that is not selected: ids up to the largest selected id index zero-length vectors (no effect),
            // larger ids are out of bounds in the reference (:141)
            if (nodes[e].chunk > max_id) panic = true;
            node_chunk[e] = 0xffffffffu;
        } else {
            node_chunk[e] = it->second;
            coverage[it->second]++;
        }
    }
    if (panic) return cc_fail(JTK_ERR_CHUNK_FAILED, "a node refers to a chunk id beyond the selected chunks");
    // ---- estimate_copy_number_of_cluster :131-181
    std::vector<uint64_t> cn_off(n_chunks + 1, 0);
    std::vector<uint32_t> cn_len(n_chunks);
    for (size_t c = 0; c < n_chunks; c++) {
        cn_len[c] = chunks[c].cluster_num;
        cn_off[c + 1] = cn_off[c] + std::max<uint32_t>(chunks[c].cluster_num, 1);
    }
    std::vector<double> cn(cn_off[n_chunks], 0.0);
    for (size_t e = 0; e < n_nodes; e++) {
        const uint32_t c = node_chunk[e];
        if (c == 0xffffffffu) continue;
        const double *p = posteriors + nodes[e].post_off;
        const double total = logsumexp(p, nodes[e].post_len);
        const size_t m = std::min<size_t>(nodes[e].post_len, cn_len[c]);
        for (size_t q = 0; q < m; q++) cn[cn_off[c] + q] += jtk_exp(p[q] - total);
    }
    for (size_t c = 0; c < n_chunks; c++) {
        double *obs = &cn[cn_off[c]];
        const size_t kk = cn_len[c], total_cp = chunks[c].copy_num;
        std::vector<double> est(kk);
        double sumf = 0.0;
        for (size_t q = 0; q < kk; q++) {
            est[q] = jtk_fmax(round_half_away(obs[q] / haploid_coverage), 1.0);
            sumf += est[q];
        }
        const size_t sum = (size_t)round_half_away(sumf);
        for (size_t it = std::min(sum, total_cp); it < total_cp; it++) {
            size_t arg = 0;
            double best = 0.0;
            bool have = false;
            for (size_t q = 0; q < kk; q++) {
                const double now = (obs[q] - est[q] * haploid_coverage) * (obs[q] - est[q] * haploid_coverage);
                const double next = (obs[q] - (est[q] + 1.0) * haploid_coverage) * (obs[q] - (est[q] + 1.0) * haploid_coverage);
                const double d = now - next;
                if (!have || !(d < best)) best = d, arg = q, have = true;
            }
            if (have) est[arg] += 1.0;
        }
        for (size_t q = 0; q < kk; q++) obs[q] = est[q];
    }
    // ---- the chunks to correct (selected_chunks order), their members (correct_chunk :191-199) and contexts (:243-261)
    struct Job {
        size_t chunk;
        std::vector<std::pair<size_t, size_t>> mem;  // (read, idx), sorted by the node's cluster (stable)
        std::vector<size_t> asn;
        size_t k = 0;
        double ari = 0.0;
        bool panic = false;
    };
    std::vector<Job> jobs;
    {
        std::vector<std::vector<std::pair<size_t, size_t>>> by_chunk(n_chunks);
        for (size_t r = 0; r < n_reads; r++)
            for (size_t idx = 0; idx < (size_t)(node_off[r + 1] - node_off[r]); idx++) {
                const uint32_t c = node_chunk[node_off[r] + idx];
                if (c != 0xffffffffu) by_chunk[c].emplace_back(r, idx);
            }
        const std::unordered_set<uint64_t> sel_set(selection, selection + n_selected);  // a genome-scale DataSet has ~1e6 chunks
        for (size_t c = 0; c < n_chunks; c++) {
            if (!(1 < chunks[c].cluster_num && sel_set.count(chunks[c].id))) continue;
            Job j;
            j.chunk = c;
            j.mem = by_chunk[c];
            std::stable_sort(j.mem.begin(), j.mem.end(), [&](const auto &a, const auto &b) {
                return nodes[node_off[a.first] + a.second].cluster < nodes[node_off[b.first] + b.second].cluster;
            });
            jobs.push_back(std::move(j));
        }
    }
    // neighbours on chunks outside `chunks` get dense ids behind the real ones (zero-length copy-number entries); the ids are
    // fixed before the first batch so that the copy-number tables are uploaded once
    std::unordered_map<uint64_t, uint32_t> stray;
    for (const Job &j : jobs)
        for (const auto &m : j.mem) {
            const size_t r = m.first, len = (size_t)(node_off[r + 1] - node_off[r]);
            for (size_t q = 0; q < len; q++)
                if (q != m.second && node_chunk[node_off[r] + q] == 0xffffffffu) {
                    const uint64_t id = nodes[node_off[r] + q].chunk;
                    if (!stray.count(id)) stray.emplace(id, (uint32_t)(n_chunks + stray.size()));
                }
        }
    const uint32_t max_dense = (uint32_t)(n_chunks + stray.size());
    std::vector<uint64_t> d_cn_off(max_dense, 0);
    std::vector<uint32_t> d_cn_len(max_dense, 0);
    for (size_t c = 0; c < n_chunks; c++) {
        d_cn_off[c] = cn_off[c];
        d_cn_len[c] = cn_len[c];
    }
    hipStream_t st = nullptr;
    std::unique_ptr<void, void (*)(void *)> st_guard(nullptr, [](void *s) { if (s) (void)hipStreamDestroy((hipStream_t)s); });
    DevBuf d_post, d_cn, d_cnoff, d_cnlen, d_panic;
    if (!jobs.empty()) {
        CC_HIP(hipStreamCreate(&st));
        st_guard.reset(st);
        size_t n_post = 0;
        for (size_t e = 0; e < n_nodes; e++) n_post = std::max<size_t>(n_post, nodes[e].post_off + nodes[e].post_len);
        std::vector<double> post_v(posteriors, posteriors + n_post);
        int rc;
        if ((rc = d_post.upload(post_v, st)) || (rc = d_cn.upload(cn, st)) || (rc = d_cnoff.upload(d_cn_off, st)) ||
            (rc = d_cnlen.upload(d_cn_len, st)))
            return cc_fail(rc, "device upload failed");
        CC_HIP(hipMalloc(&d_panic.p, sizeof(int)));
        CC_HIP(hipMemsetAsync(d_panic.p, 0, sizeof(int), st));
        CC_HIP(hipStreamSynchronize(st));  // post_v goes out of scope
    }
    g_first_sims.clear();
    // The reference corrects one chunk at a time under rayon (phmm_likelihood_correction.rs:37-43).  Here the jobs go through the
    // device in BATCHES bounded by the bytes of their similarity matrices (sum of n^2 doubles), so host and device memory stay
    // bounded on a genome-scale DataSet; a batch is: similarity fill on the device, then the spectral step on host threads.
    uint64_t sims_budget = 64ull << 20;  // doubles: 512 MB per batch
    if (const char *e = getenv("JTK_CC_SIMS_BUDGET")) {  // tests force several batches on a small problem
        const long long v = atoll(e);
        if (v > 0) sims_budget = (uint64_t)v;
    }
    for (size_t j0 = 0, j1 = 0; j0 < jobs.size(); j0 = j1) {
    uint64_t batch_sims = 0;
    for (j1 = j0; j1 < jobs.size(); j1++) {
        const uint64_t nn = (uint64_t)jobs[j1].mem.size() * jobs[j1].mem.size();
        if (j1 > j0 && batch_sims + nn > sims_budget) break;
        batch_sims += nn;
    }
    std::vector<ArmEnt> arms;
    std::vector<Member> members;
    std::vector<PairJob> pjobs;
    uint64_t n_pairs = 0, sims_total = 0;
    uint32_t max_arm = 0;
    for (size_t jq = j0; jq < j1; jq++) {
        const Job &j = jobs[jq];
        PairJob pj;
        pj.first_pair = n_pairs;
        pj.sims_off = sims_total;
        pj.member0 = (uint32_t)members.size();
        pj.n = (uint32_t)j.mem.size();
        pj.chunk = (uint32_t)j.chunk;
        pj.pad = 0;
        pjobs.push_back(pj);
        n_pairs += (uint64_t)pj.n * pj.n;
        sims_total += (uint64_t)pj.n * pj.n;
        for (const auto &m : j.mem) {
            const size_t r = m.first, idx = m.second, len = (size_t)(node_off[r + 1] - node_off[r]);
            const jtk_cc_node_t *rn = nodes + node_off[r];
            auto ent = [&](size_t q) {
                ArmEnt e;
                uint32_t c = node_chunk[node_off[r] + q];
                if (c == 0xffffffffu) {
                    // a neighbour on a chunk outside `chunks`: only ever compared with the same chunk id, whose copy numbers the
                    // reference reads from a zero-length vector -> sim() asserts the lengths: the kernel sees cn_len 0 too
                    c = stray.at(rn[q].chunk);
                }
                e.chunk = c;
                e.post_len = rn[q].post_len;
                e.post_off = rn[q].post_off;
                return e;
            };
            Member mb;
            std::vector<ArmEnt> before, after;
            for (size_t q = 0; q < idx; q++) before.push_back(ent(idx - 1 - q));  // nodes[..idx] reversed
            for (size_t q = idx + 1; q < len; q++) after.push_back(ent(q));
            const bool fwd = rn[idx].is_forward != 0;
            const std::vector<ArmEnt> &up = fwd ? before : after, &down = fwd ? after : before;
            mb.up_off = arms.size();
            mb.n_up = (uint32_t)up.size();
            arms.insert(arms.end(), up.begin(), up.end());
            mb.down_off = arms.size();
            mb.n_down = (uint32_t)down.size();
            arms.insert(arms.end(), down.begin(), down.end());
            mb.post_off = rn[idx].post_off;
            mb.post_len = rn[idx].post_len;
            mb.pad = 0;
            members.push_back(mb);
            max_arm = std::max(max_arm, std::max(mb.n_up, mb.n_down));
        }
    }
    // ---- the similarity fill on the device
    std::vector<double> sims(sims_total);
    if (n_pairs) {
        DevBuf d_jobs, d_members, d_arms, d_sims, d_scratch;
        int rc;
        if ((rc = d_jobs.upload(pjobs, st)) || (rc = d_members.upload(members, st)) || (rc = d_arms.upload(arms, st)))
            return cc_fail(rc, "device upload failed");
        CC_HIP(hipMalloc(&d_sims.p, std::max<size_t>(sims_total, 1) * sizeof(double)));
        const uint32_t row_doubles = 2 * 3 * (max_arm + 1);
        uint64_t threads = std::min<uint64_t>(n_pairs, 256ull * 1024);
        threads = (threads + 255) / 256 * 256;
        CC_HIP(hipMalloc(&d_scratch.p, threads * row_doubles * sizeof(double)));
        similarity_kernel<<<(uint32_t)(threads / 256), 256, 0, st>>>(
            n_pairs, (uint32_t)pjobs.size(), (const PairJob *)d_jobs.p, (const Member *)d_members.p, (const ArmEnt *)d_arms.p,
            (const double *)d_post.p, (const double *)d_cn.p, (const uint64_t *)d_cnoff.p, (const uint32_t *)d_cnlen.p,
            (double *)d_sims.p, (double *)d_scratch.p, row_doubles, (int *)d_panic.p);
        int dev_panic = 0;
        CC_HIP(hipMemcpyAsync(sims.data(), d_sims.p, sims_total * sizeof(double), hipMemcpyDeviceToHost, st));
        CC_HIP(hipMemcpyAsync(&dev_panic, d_panic.p, sizeof(int), hipMemcpyDeviceToHost, st));
        CC_HIP(hipStreamSynchronize(st));
        CC_HIP(hipGetLastError());
        if (dev_panic) return cc_fail(JTK_ERR_CHUNK_FAILED, "sim(): posterior lengths differ from cluster_num, or a log-probability above 0");
    }
    if (g_keep_sims && j0 == 0 && !pjobs.empty()) g_first_sims.assign(sims.begin(), sims.begin() + (size_t)pjobs[0].n * pjobs[0].n);
    // ---- spectral clustering of every chunk (clustering :290-337), one chunk per host thread
    auto cluster_one = [&](size_t ji) {
        Job &j = jobs[ji];
        const jtk_cc_chunk_t &chunk = chunks[j.chunk];
        const size_t n = j.mem.size();
        double *S = &sims[pjobs[ji - j0].sims_off];
        if (chunk.copy_num == 0 || n == 0) {
            j.panic = true;
            return;
        }
        const size_t len = n - n / chunk.copy_num / 4;  // cov_per_copy
        if (!(len < n)) {  // select_nth indexes sims[pivot] with pivot == len (:361)
            j.panic = true;
            return;
        }
        {  // filter_similarity :330-347
            std::vector<uint8_t> keep(n * n, 0);
            std::vector<double> tmp(n);
            for (size_t i = 0; i < n; i++) {
                std::copy(S + i * n, S + (i + 1) * n, tmp.begin());
                std::sort(tmp.begin(), tmp.end());
                const double threshold = jtk_fmax(tmp[len], 0.51);
                for (size_t q = 0; q < n; q++)
                    if (threshold <= S[i * n + q]) keep[i * n + q] = keep[q * n + i] = 1;
            }
            for (size_t e = 0; e < n * n; e++)
                if (!keep[e]) S[e] = 0.0000000000000001;
        }
        std::vector<double> rowsum(n), sq_inv(n), lap(n * n), vec(n * n);
        for (size_t i = 0; i < n; i++) {  // get_graph_laplacian :385-402
            double s = 0.0;
            for (size_t q = 0; q < n; q++) s += S[i * n + q];
            rowsum[i] = s;
            sq_inv[i] = std::sqrt(1.0 / s);
        }
        for (size_t i = 0; i < n; i++)
            for (size_t q = 0; q < n; q++) lap[i * n + q] = q == i ? 1.0 : -S[i * n + q] * sq_inv[i] * sq_inv[q];
        jtk_symmetric_eigen(lap.data(), n, vec.data());  // get_eigenvalues :405-464
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return std::fabs(lap[a * n + a]) < std::fabs(lap[b * n + b]); });
        size_t pick_k = 0;
        while (pick_k < n && lap[order[pick_k] * n + order[pick_k]] < 0.2) pick_k++;
        if (pick_k == 0) {
            j.panic = true;
            return;
        }
        const size_t pl = chunk.cluster_num, dim = pick_k + pl;
        std::vector<double> feat(n * dim);
        for (size_t i = 0; i < n; i++) {
            const double d = std::sqrt(1.0 / rowsum[i]);
            for (size_t q = 0; q < pick_k; q++) feat[i * dim + q] = vec[i * n + order[q]] * d;
            const jtk_cc_node_t &nd = nodes[node_off[j.mem[i].first] + j.mem[i].second];
            if (nd.post_len != pl) {
                j.panic = true;
                return;
            }
            const double *p = posteriors + nd.post_off;
            const double total = logsumexp(p, pl);
            for (size_t q = 0; q < pl; q++) feat[i * dim + pick_k + q] = jtk_exp(p[q] - total);
        }
        for (size_t q = 0; q < dim; q++) {  // normalize_columns :369-381
            double s = 0.0;
            for (size_t i = 0; i < n; i++) s += feat[i * dim + q] * feat[i * dim + q];
            s = std::sqrt(s);
            for (size_t i = 0; i < n; i++) feat[i * dim + q] /= s;
        }
        Rng128 rng(chunk.id * (uint64_t)chunk.cluster_num);  // :299-301
        const size_t cluster_num = std::min<size_t>(chunk.cluster_num, pick_k);
        std::vector<size_t> cur;
        double best = 0.0;
        bool have = false;
        for (int it = 0; it < 20; it++) {  // :303-307: min_by keeps the first minimum
            double dist = 0.0;
            if (!kmeans(feat, n, dim, cluster_num, rng, &dist, cur)) {
                j.panic = true;
                return;
            }
            if (!have || dist < best) {
                best = dist;
                have = true;
                j.asn = cur;
            }
        }
        j.k = cluster_num;
        // adj_rand_on_biased :220-240
        std::vector<size_t> prev(n), pb, ab;
        for (size_t i = 0; i < n; i++) {
            const jtk_cc_node_t &nd = nodes[node_off[j.mem[i].first] + j.mem[i].second];
            prev[i] = (size_t)nd.cluster;
            const double *p = posteriors + nd.post_off;
            bool biased = nd.post_len <= 1;  // Node::is_biased, definitions/src/lib.rs:703-709
            const double thr = 1.0 / (double)nd.post_len + 0.2;
            for (size_t q = 0; q < nd.post_len && !biased; q++) biased = thr <= jtk_exp(p[q]);
            if (biased) {
                pb.push_back((size_t)nd.cluster);
                ab.push_back(j.asn[i]);
            }
        }
        bool pnc = false;
        (void)adjusted_rand_index(prev, j.asn, &pnc);
        const double adj = adjusted_rand_index(pb, ab, &pnc);
        j.panic = pnc;
        j.ari = adj != adj ? 1.0 : adj;
    };
    {
        std::atomic<size_t> next(j0);
        auto work = [&]() {
            for (size_t ji = next.fetch_add(1); ji < j1; ji = next.fetch_add(1)) cluster_one(ji);
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nt = std::min<size_t>(std::max<size_t>(j1 - j0, 1), std::min<size_t>(hw ? hw : 1, 32));
        std::vector<std::thread> threads;
        for (size_t t = 1; t < nt; t++) threads.emplace_back(work);
        work();
        for (auto &t : threads) t.join();
    }
    }  // batches
    for (const Job &j : jobs)
        if (j.panic) return cc_fail(JTK_ERR_CHUNK_FAILED, "chunk " + std::to_string(chunks[j.chunk].id) + ": the reference panics on this pile-up");
    // ---- get_protected_clusterings :108-129, supress_threshold :100-105, write-back :46-96
    std::vector<uint8_t> prot(n_chunks, 0);
    for (size_t c = 0; c < n_chunks; c++) {
        if (coverage[c] == 0) continue;
        const double cl = (double)chunks[c].cluster_num, improve_frac = (cl - 1.0) / cl;
        prot[c] = (double)coverage[c] * improve_frac * min_gain < chunks[c].score;
    }
    std::vector<double> aris;
    for (const Job &j : jobs) aris.push_back(j.ari);
    std::sort(aris.begin(), aris.end());
    const size_t pick = (size_t)std::ceil((double)aris.size() * 0.05);
    const double supress_cluster = pick < aris.size() ? aris[pick] : 1.0;
    for (const Job &j : jobs)
        if (!(j.k <= chunks[j.chunk].copy_num)) return cc_fail(JTK_ERR_CHUNK_FAILED, "cluster_num above copy_num");  // assert! :55
    for (const Job &j : jobs) {
        jtk_cc_chunk_t &chunk = chunks[j.chunk];
        const bool supress = j.k == 1 || j.ari < supress_cluster;
        if (supress && prot[j.chunk]) continue;
        chunk.cluster_num = supress ? 1 : (uint32_t)j.k;
        for (size_t m = 0; m < j.mem.size(); m++) {
            const size_t e = (size_t)node_off[j.mem[m].first] + j.mem[m].second;
            cluster_out[e] = supress ? 0 : (uint64_t)j.asn[m];
            touched[e] = 1;
        }
    }
    return 0;
}
