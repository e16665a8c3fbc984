// mcmc_kernels.hip -- the read-clustering loop on the device: one wavefront per chunk.
//
// Follows, statement by statement, haplotyper/src/local_clustering/pseudo_mcmc.rs
//   cluster_filtered_variants :213-274   mcmc_clustering :649-670   mcmc_with_filter :704-762
//   flip :764-783   get_lk :785-795   LKCount :797-845   get_used_columns :847-869
//   get_read_lk_gains :381-408   get_likelihood_gain :353-379   use_highest_gain :673-693
//   expected_gains :286-306   clustering tail :98-105   to_posterior_probability :342-347
// and haplotyper/src/misc.rs  kmeans :231-259, suggest_first :315-341, logsumexp :84-92,
// with the rand 0.8.5 / rand_xoshiro 0.6.0 sampling restated exactly as in oracle/rng.c.
//
// The chain is strictly sequential per chunk (one RNG stream threads through k-means, 20 restarts and every
// candidate k, local_clustering/mod.rs:97), so the parallelism is: chunks across wavefronts, and inside a
// step the D variant columns across lanes.  Lane d keeps LKCount[c][d] for every cluster c in registers
// (K is a template parameter so the cluster index is a static register index); the RNG state and the
// cluster sizes are wave-uniform.  Sums that the reference evaluates left to right are evaluated left to
// right here (serial chains over LDS broadcast reads) -- integer labels only match if every f64 rounding
// matches.
#include "device_common.h"

namespace {

struct Rng {
    uint64_t s0, s1, s2, s3;
};
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
__device__ __forceinline__ uint64_t splitmix64(uint64_t &x) {
    x += 0x9e3779b97f4a7c15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t next_u64(Rng &r) {
    const uint64_t result = rotl64(r.s1 * 5, 7) * 9;
    const uint64_t t = r.s1 << 17;
    r.s2 ^= r.s0;
    r.s3 ^= r.s1;
    r.s1 ^= r.s2;
    r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = rotl64(r.s3, 45);
    return result;
}
__device__ __forceinline__ uint32_t next_u32(Rng &r) { return (uint32_t)(next_u64(r) >> 32); }
__device__ __forceinline__ uint64_t gen_range_usize(Rng &r, uint64_t n) {
    const uint64_t zone = (n << __clzll((long long)n)) - 1;
    for (;;) {
        const uint64_t v = next_u64(r);
        const uint64_t hi = __umul64hi(v, n), lo = v * n;
        if (lo <= zone) return hi;
    }
}
__device__ __forceinline__ uint32_t gen_range_u32(Rng &r, uint32_t n) {
    const uint32_t zone = (n << __clz((int)n)) - 1;
    for (;;) {
        const uint32_t v = next_u32(r);
        const uint64_t m = (uint64_t)v * n;
        if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
    }
}
__device__ __forceinline__ uint64_t gen_index(Rng &r, uint64_t ub) {
    return ub <= 0xffffffffULL ? gen_range_u32(r, (uint32_t)ub) : gen_range_usize(r, ub);
}
__device__ __forceinline__ bool gen_bool(Rng &r, double p) {
    if (p == 1.0) return true;
    const double scaled = p * 18446744073709551616.0;
    const uint64_t p_int = !(scaled > 0.0) ? 0ull : __double2ull_rz(scaled);
    return next_u64(r) < p_int;
}
__device__ __forceinline__ uint32_t choose_other(Rng &r, uint32_t k, uint32_t old) {
    uint32_t result = 0xffffffffu, consumed = 0;
    for (uint32_t c = 0; c < k; c++) {
        if (c == old) continue;
        consumed++;
        if (gen_index(r, consumed) == 0) result = c;
    }
    return result;
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS work area of one chunk
struct Lds {
    double *data;        // n x D
    double *size_to_lk;  // n + 1
    double *lfact;       // n + 1
    double *val;         // K x D staging of the per-(cluster, column) terms
    double *centers;     // K x D
    double *fbuf;        // n (dists / weights / per-read gains)
    double *cum;         // n
    uint8_t *thr;        // n + 1 : min num_pos that makes a column informative at num_pos+num_neg = s
    uint8_t *assign;     // n   current labels
    uint8_t *argmax;     // n   best labels seen in this chain
    uint8_t *best;       // n   best over restarts for this k
    uint8_t *accepted;   // n   labels of the accepted k
    uint8_t *used;       // D
    uint8_t *prev_used;  // D
    uint8_t *tmp_asn;    // n
    uint8_t *tmp_used;   // D
};

// slice.choose_weighted over weights w[0..n) in LDS; cum is scratch. Returns -1 on WeightedError.
__device__ int choose_weighted(Rng &r, const double *w, uint32_t n, double *cum, uint32_t lane) {
    double total = w[0];
    if (!(total >= 0.0)) return -1;
    bool bad = false;
    for (uint32_t i = 1; i < n; i++) {
        const double wi = w[i];
        if (!(wi >= 0.0)) bad = true;
        if (lane == 0) cum[i - 1] = total;
        total += wi;
    }
    if (bad || total == 0.0) return -1;
    double scale = total;
    const double max_rand = 1.0 - 0x1p-52;
    while (scale * max_rand + 0.0 >= total) scale = jtk_bits_f64(jtk_f64_bits(scale) - 1);
    const double v12 = jtk_bits_f64((next_u64(r) >> 12) | 0x3ff0000000000000ULL);
    const double chosen = (v12 - 1.0) * scale + 0.0;
    __syncthreads();
    // partition point of `cum[i] <= chosen` (cum is non-decreasing): count the entries <= chosen
    uint32_t cnt = 0;
    for (uint32_t i = lane; i + 1 < n; i += 64) cnt += cum[i] <= chosen ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    __syncthreads();
    return (int)cnt;
}

__device__ __forceinline__ double dist_row(const double *a, const double *b, uint32_t D) {
    double s = 0.0;
    for (uint32_t d = 0; d < D; d++) {
        const double t = a[d] - b[d];
        s += t * t;
    }
    return s;
}

// misc.rs:261-276 with centres given as K rows of D doubles in LDS (first minimum wins)
__device__ void update_assignments(const Lds &m, uint32_t n, uint32_t D, uint32_t k, const double *centers,
                                   uint8_t *assign, uint32_t lane) {
    for (uint32_t i = lane; i < n; i += 64) {
        uint32_t best = 0;
        double bd = dist_row(m.data + i * D, centers, D);
        for (uint32_t c = 1; c < k; c++) {
            const double d = dist_row(m.data + i * D, centers + c * D, D);
            if (d < bd) {
                bd = d;
                best = c;
            }
        }
        assign[i] = (uint8_t)best;
    }
    __syncthreads();
}

// misc.rs:298-307: sum over reads, in read order, of dist(read, its centre)
__device__ double get_dist(const Lds &m, uint32_t n, uint32_t D, const uint8_t *assign, uint32_t lane) {
    for (uint32_t i = lane; i < n; i += 64) m.fbuf[i] = dist_row(m.data + i * D, m.centers + assign[i] * D, D);
    __syncthreads();
    double s = 0.0;
    for (uint32_t i = 0; i < n; i++) s += m.fbuf[i];
    __syncthreads();
    return s;
}

// misc.rs:229-259; returns false where the reference would panic
__device__ bool kmeans(const Lds &m, uint32_t n, uint32_t D, uint32_t k, Rng &rng, uint32_t lane) {
    const double UPDATE_THR = 0.00000001;
    if (gen_bool(rng, 0.5)) {
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t c = (uint32_t)gen_range_usize(rng, k);
            if (lane == 0) m.assign[i] = (uint8_t)c;
        }
        __syncthreads();
    } else {
        // suggest_first (misc.rs:315-341): centre rows are borrowed data rows; keep their indices in cum's tail
        uint32_t centre_idx[JTK_MAX_COPY];
        centre_idx[0] = (uint32_t)gen_index(rng, n);
        uint32_t nc = 1;
        for (uint32_t it = 0; it + 1 < k; it++) {
            for (uint32_t i = lane; i < n; i += 64) {
                double mn = dist_row(m.data + i * D, m.data + centre_idx[0] * D, D);
                for (uint32_t c = 1; c < nc; c++) {
                    const double d = dist_row(m.data + i * D, m.data + centre_idx[c] * D, D);
                    if (d < mn) mn = d;
                }
                m.fbuf[i] = mn;
            }
            __syncthreads();
            const int idx = choose_weighted(rng, m.fbuf, n, m.cum, lane);
            if (idx < 0) return false;
            centre_idx[nc++] = (uint32_t)idx;
        }
        for (uint32_t c = 0; c < k; c++)
            for (uint32_t d = lane; d < D; d += 64) m.centers[c * D + d] = m.data[centre_idx[c] * D + d];
        __syncthreads();
        update_assignments(m, n, D, k, m.centers, m.assign, lane);
    }
    // Lloyd iterations; `dist` is first evaluated against all-zero centres
    for (uint32_t e = lane; e < k * D; e += 64) m.centers[e] = 0.0;
    __syncthreads();
    double dist = get_dist(m, n, D, m.assign, lane);
    for (;;) {
        // update_centers (misc.rs:277-297): per (cluster, column) slot, sum in read order
        for (uint32_t e = lane; e < k * D; e += 64) {
            const uint32_t c = e / D, d = e % D;
            double s = 0.0;
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < n; i++)
                if (m.assign[i] == c) {
                    s += m.data[i * D + d];
                    cnt++;
                }
            m.centers[e] = cnt > 0 ? s / (double)cnt : s;
        }
        __syncthreads();
        update_assignments(m, n, D, k, m.centers, m.assign, lane);
        const double nd = get_dist(m, n, D, m.assign, lane);
        if (!(nd < dist + UPDATE_THR)) return false;  // assert!(new_dist < dist + UPDATE_THR)
        if (dist - nd < UPDATE_THR) break;
        dist = nd;
    }
    return true;
}

// Per-lane LKCount of one column for K clusters.
template <int K>
struct Counts {
    double tg[K];
    int np[K], nn[K];
};

template <int K>
__device__ __forceinline__ void lk_add(Counts<K> &q, uint32_t c, double x) {
#pragma unroll
    for (int cc = 0; cc < K; cc++)
        if ((uint32_t)cc == c) {  // c is wave-uniform: a scalar branch, static register index
            q.tg[cc] += x;
            if (JTK_POS_THR < x)
                q.np[cc]++;
            else if (x < -JTK_POS_THR)
                q.nn[cc]++;
        }
}
template <int K>
__device__ __forceinline__ void lk_sub(Counts<K> &q, uint32_t c, double x) {
#pragma unroll
    for (int cc = 0; cc < K; cc++)
        if ((uint32_t)cc == c) {
            q.tg[cc] -= x;
            if (JTK_POS_THR < x)
                q.np[cc]--;
            else if (x < -JTK_POS_THR)
                q.nn[cc]--;
        }
}

// get_used_columns (:847-869) for this lane's column
template <int K>
__device__ __forceinline__ bool column_used(const Counts<K> &q, const uint8_t *thr) {
    bool any = false;
    int in_use = 0, in_neg = 0;
#pragma unroll
    for (int c = 0; c < K; c++) {
        const bool pos = 0.0 < q.tg[c];
        // is_informative: 0 < total_gain && 0.70 < num_pos / (num_pos + num_neg + 1e-7)
        any |= pos && q.np[c] >= (int)thr[q.np[c] + q.nn[c]];
        in_use += pos ? q.np[c] : 0;
        in_neg += pos ? 0 : q.np[c];
    }
    return any && 2 * in_neg < in_use;
}

template <int K>
__device__ __forceinline__ void fill_counts(const Lds &m, uint32_t n, uint32_t D, const uint8_t *assign,
                                            Counts<K> &q, int *clusters, uint32_t lane) {
#pragma unroll
    for (int c = 0; c < K; c++) {
        q.tg[c] = 0.0;
        q.np[c] = 0;
        q.nn[c] = 0;
        clusters[c] = 0;
    }
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = uni(assign[i]);
        const double x = lane < D ? m.data[i * D + lane] : 0.0;
        lk_add<K>(q, c, x);
#pragma unroll
        for (int cc = 0; cc < K; cc++)
            if ((uint32_t)cc == c) clusters[cc]++;
    }
}

// get_lk (:785-795): size terms first, then clusters outer / columns inner, left to right
template <int K>
__device__ __forceinline__ double get_lk(const Lds &m, const Counts<K> &q, const int *clusters, uint32_t D,
                                         uint32_t lane) {
    const bool used = lane < D && column_used<K>(q, m.thr);
#pragma unroll
    for (int c = 0; c < K; c++)
        if (lane < D) m.val[c * D + lane] = used ? jtk_fmax(q.tg[c], 0.0) : 0.0;
    __syncthreads();
    double lk = 0.0;
#pragma unroll
    for (int c = 0; c < K; c++) lk += m.size_to_lk[clusters[c]];
    const uint32_t tot = K * D;
    for (uint32_t s = 0; s < tot; s++) lk += m.val[s];
    __syncthreads();
    return lk;
}

template <int K>
__device__ __forceinline__ void flip(const Lds &m, Counts<K> &q, int *clusters, uint32_t D, uint32_t idx,
                                     uint32_t from, uint32_t to, uint32_t lane) {
    const double x = lane < D ? m.data[idx * D + lane] : 0.0;
    lk_sub<K>(q, from, x);
    lk_add<K>(q, to, x);
#pragma unroll
    for (int cc = 0; cc < K; cc++) {
        if ((uint32_t)cc == from) clusters[cc]--;
        if ((uint32_t)cc == to) clusters[cc]++;
    }
    if (lane == 0) m.assign[idx] = (uint8_t)to;
}

// mcmc_with_filter (:704-762). m.assign holds the k-means labels on entry, the best-seen labels on exit.
template <int K>
__device__ double mcmc_with_filter(const Lds &m, uint32_t n, uint32_t D, double cov, Rng &rng, uint32_t lane) {
    // size_to_lk[x] = max_{c=1..K} poisson_lk(x, cov*c)
    for (uint32_t x = lane; x <= n; x += 64) {
        double mx = -__builtin_inf();
        for (int c = 1; c <= K; c++) {
            const double lam = cov * (double)c;
            mx = jtk_fmax(mx, (double)x * jtk_log(lam) - lam - m.lfact[x]);
        }
        m.size_to_lk[x] = mx;
    }
    __syncthreads();
    Counts<K> q;
    int clusters[K];
    fill_counts<K>(m, n, D, m.assign, q, clusters, lane);
    double lk = get_lk<K>(m, q, clusters, D, lane);
    double max = lk;
    for (uint32_t i = lane; i < n; i += 64) m.argmax[i] = m.assign[i];
    __syncthreads();
    const uint32_t total = 2000u * n;
    for (uint32_t t = 0; t < total; t++) {
        const uint32_t idx = (uint32_t)gen_range_usize(rng, n);
        const uint32_t old = uni(m.assign[idx]);
        const uint32_t nw = choose_other(rng, K, old);
        flip<K>(m, q, clusters, D, idx, old, nw, lane);
        const double proposed = get_lk<K>(m, q, clusters, D, lane);
        const double diff = proposed - lk;
        if (0.0 < diff || gen_bool(rng, jtk_exp(diff))) {
            lk = proposed;
            if (max < lk) {
                max = proposed;
                for (uint32_t i = lane; i < n; i += 64) m.argmax[i] = m.assign[i];
                __syncthreads();
            }
        } else {
            flip<K>(m, q, clusters, D, idx, nw, old, lane);
            __syncthreads();
        }
    }
    for (uint32_t i = lane; i < n; i += 64) m.assign[i] = m.argmax[i];
    __syncthreads();
    return max;
}

// get_read_lk_gains (:381-408): used columns -> used[], per-read gain -> fbuf[]
template <int K>
__device__ void get_read_lk_gains(const Lds &m, uint32_t n, uint32_t D, const uint8_t *assign, uint8_t *used,
                                  uint32_t lane) {
    Counts<K> q;
    int clusters[K];
    fill_counts<K>(m, n, D, assign, q, clusters, lane);
    const bool u = lane < D && column_used<K>(q, m.thr);
    if (lane < D) used[lane] = u ? 1 : 0;
#pragma unroll
    for (int c = 0; c < K; c++)
        if (lane < D) m.val[c * D + lane] = (u && JTK_POS_THR < q.tg[c]) ? 1.0 : 0.0;  // column counts for cluster c
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t a = assign[i];
        double s = 0.0;
        for (uint32_t d = 0; d < D; d++)
            if (m.val[a * D + d] != 0.0) s += m.data[i * D + d];
        m.fbuf[i] = s;
    }
    __syncthreads();
}

// get_likelihood_gain (:353-379): out[i*K + c]
template <int K>
__device__ void get_likelihood_gain(const Lds &m, uint32_t n, uint32_t D, const uint8_t *assign, double *out,
                                    uint32_t lane) {
    Counts<K> q;
    int clusters[K];
    fill_counts<K>(m, n, D, assign, q, clusters, lane);
    const bool u = lane < D && column_used<K>(q, m.thr);
#pragma unroll
    for (int c = 0; c < K; c++)
        if (lane < D) m.val[c * D + lane] = (u && JTK_POS_THR < q.tg[c]) ? 1.0 : 0.0;
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64)
        for (int c = 0; c < K; c++) {
            double s = 0.0;
            for (uint32_t d = 0; d < D; d++)
                if (m.val[c * D + d] != 0.0) s += m.data[i * D + d];
            out[i * K + c] = s;
        }
    __syncthreads();
}

// mcmc_clustering (:649-670): labels -> m.best, per-read gains -> m.fbuf, used columns -> m.used
template <int K>
__device__ bool mcmc_clustering(const Lds &m, uint32_t n, uint32_t D, double cov, Rng &rng, double *score,
                                uint32_t lane) {
    double best = 0.0;
    bool have = false;
    for (int it = 0; it < 20; it++) {
        if (!kmeans(m, n, D, K, rng, lane)) return false;
        const double lk = mcmc_with_filter<K>(m, n, D, cov, rng, lane);
        if (!have || !(lk < best)) {  // max_by: the last maximum wins
            best = lk;
            have = true;
            for (uint32_t i = lane; i < n; i += 64) m.best[i] = m.assign[i];
            __syncthreads();
        }
    }
    get_read_lk_gains<K>(m, n, D, m.best, m.used, lane);
    // cluster_lk = sum_c max_poisson_lk(count_c, cov, 1, K)
    double cluster_lk = 0.0;
    for (int c = 0; c < K; c++) {
        uint32_t cnt = 0;
        for (uint32_t i = 0; i < n; i++) cnt += m.best[i] == c ? 1u : 0u;
        double mx = -__builtin_inf();
        for (int cc = 1; cc <= K; cc++) {
            const double lam = cov * (double)cc;
            mx = jtk_fmax(mx, (double)cnt * jtk_log(lam) - lam - m.lfact[cnt]);
        }
        cluster_lk += mx;
    }
    *score = best - cluster_lk;
    return true;
}

template <int K>
__device__ bool run_k(const Lds &m, uint32_t n, uint32_t D, double cov, Rng &rng, double *score, uint32_t lane) {
    return mcmc_clustering<K>(m, n, D, cov, rng, score, lane);
}

__device__ bool run_k_dyn(uint32_t k, const Lds &m, uint32_t n, uint32_t D, double cov, Rng &rng, double *score,
                          uint32_t lane) {
    switch (k) {
        case 2: return run_k<2>(m, n, D, cov, rng, score, lane);
        case 3: return run_k<3>(m, n, D, cov, rng, score, lane);
        case 4: return run_k<4>(m, n, D, cov, rng, score, lane);
        case 5: return run_k<5>(m, n, D, cov, rng, score, lane);
        case 6: return run_k<6>(m, n, D, cov, rng, score, lane);
        case 7: return run_k<7>(m, n, D, cov, rng, score, lane);
        default: return false;
    }
}

__device__ void likelihood_gain_dyn(uint32_t k, const Lds &m, uint32_t n, uint32_t D, const uint8_t *assign,
                                    double *out, uint32_t lane) {
    switch (k) {
        case 1: get_likelihood_gain<1>(m, n, D, assign, out, lane); break;
        case 2: get_likelihood_gain<2>(m, n, D, assign, out, lane); break;
        case 3: get_likelihood_gain<3>(m, n, D, assign, out, lane); break;
        case 4: get_likelihood_gain<4>(m, n, D, assign, out, lane); break;
        case 5: get_likelihood_gain<5>(m, n, D, assign, out, lane); break;
        case 6: get_likelihood_gain<6>(m, n, D, assign, out, lane); break;
        default: get_likelihood_gain<7>(m, n, D, assign, out, lane); break;
    }
}

__device__ __forceinline__ double gains_expected(const jtk_gains_t *g, uint32_t homop_len, int dt) {
    if (homop_len == 0) homop_len = 1;
    const uint32_t h = homop_len < g->max_homopolymer_len ? homop_len : g->max_homopolymer_len;
    return dt == JTK_DIFF_SUBST ? g->subst[h - 1].gain
                                : (dt == JTK_DIFF_DEL ? g->deletions[h - 1].gain : g->insertions[h - 1].gain);
}

// one wave per chunk
__global__ __launch_bounds__(64) void mcmc_kernel(const ChunkMeta *chunks, ChunkState *state,
                                                  const jtk_lc_params_t *params, const double *feat_all,
                                                  const uint32_t *vtype_all, const uint64_t *vt_off_all,
                                                  uint32_t vt_stride_mode, uint32_t *label_all, double *post_all,
                                                  uint32_t post_stride, double *lg_all, const uint64_t *lg_off,
                                                  uint32_t lds_n, uint32_t lds_d) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t ci = blockIdx.x, lane = threadIdx.x;
    ChunkState *st = &state[ci];
    if (st->status != 0) return;
    const ChunkMeta cm = chunks[ci];
    const uint32_t n = cm.n_reads, D = st->dim, copy_num = cm.copy_num;
    const double coverage = params->haploid_coverage;
    uint32_t *label = label_all + cm.read_first;
    double *post = post_all + (uint64_t)cm.read_first * post_stride;
    // ---- trivial outcomes (pseudo_mcmc.rs:86-88, :221-225)
    if (copy_num < 2 || D == 0 || n <= copy_num) {
        for (uint32_t i = lane; i < n; i += 64) {
            label[i] = 0;
            for (uint32_t c = 0; c < post_stride; c++) post[(uint64_t)i * post_stride + c] = 0.0;
        }
        if (lane == 0) {
            st->score = 0.0;
            st->k = 1;
        }
        return;
    }
    if (copy_num > JTK_MAX_COPY || n > 255 || n > lds_n || D > lds_d) {
        if (lane == 0) st->status = JTK_ERR_UNSUPPORTED;
        return;
    }
    // ---- LDS carve
    Lds m;
    {
        unsigned char *p = smem;
        auto take = [&](size_t bytes) {
            unsigned char *q = p;
            p += (bytes + 15) & ~(size_t)15;
            return q;
        };
        m.data = (double *)take((size_t)lds_n * lds_d * 8);
        m.size_to_lk = (double *)take((size_t)(lds_n + 1) * 8);
        m.lfact = (double *)take((size_t)(lds_n + 1) * 8);
        m.val = (double *)take((size_t)JTK_MAX_COPY * lds_d * 8);
        m.centers = (double *)take((size_t)JTK_MAX_COPY * lds_d * 8);
        m.fbuf = (double *)take((size_t)lds_n * 8);
        m.cum = (double *)take((size_t)lds_n * 8);
        m.thr = (uint8_t *)take(lds_n + 1);
        m.assign = (uint8_t *)take(lds_n);
        m.argmax = (uint8_t *)take(lds_n);
        m.best = (uint8_t *)take(lds_n);
        m.accepted = (uint8_t *)take(lds_n);
        m.tmp_asn = (uint8_t *)take(lds_n);
        m.used = (uint8_t *)take(lds_d);
        m.prev_used = (uint8_t *)take(lds_d);
        m.tmp_used = (uint8_t *)take(lds_d);
    }
    const double *feat = feat_all + cm.feat_off;
    for (uint32_t e = lane; e < n * D; e += 64) m.data[e] = feat[e];
    // lfact[x] = sum_{c=1..x} ln c, summed left to right as poisson_lk does (:636-638)
    if (lane == 0) {
        double s = 0.0;
        m.lfact[0] = 0.0;
        for (uint32_t c = 1; c <= n; c++) {
            s += jtk_log((double)c);
            m.lfact[c] = s;
        }
    }
    // thr[s]: smallest num_pos with 0.70 < num_pos / (s + 1e-7)  (LKCount::is_informative, :818-822);
    // the quotient is monotone in num_pos for fixed s, so the test is num_pos >= thr[s]
    for (uint32_t s = lane; s <= n; s += 64) {
        const double cov = (double)s + 0.0000001;
        uint32_t t = s + 1;  // "never"
        for (uint32_t p = 0; p <= s; p++)
            if (0.70 < (double)p / cov) {
                t = p;
                break;
            }
        m.thr[s] = (uint8_t)(t > 255 ? 255 : t);
    }
    for (uint32_t d = lane; d < D; d += 64) m.prev_used[d] = 0;
    for (uint32_t i = lane; i < n; i += 64) m.accepted[i] = 0;
    __syncthreads();
    const uint32_t *vt = vtype_all + 2 * ((uint64_t)ci * JTK_MAX_DIM);
    if (vt_stride_mode) vt = vtype_all + 2 * vt_off_all[ci];
    // ---- per-chunk RNG (local_clustering/mod.rs:97)
    Rng rng;
    {
        uint64_t x = cm.chunk_id * 3490ULL;
        rng.s0 = splitmix64(x);
        rng.s1 = splitmix64(x);
        rng.s2 = splitmix64(x);
        rng.s3 = splitmix64(x);
    }
    // ---- cluster_filtered_variants (:213-274)
    const double per_cluster_cov = cm.local_coverage;
    double max = 0.0;
    uint32_t max_k = 1;
    const uint32_t end = copy_num < 1 + 2 * D ? copy_num : 1 + 2 * D;
    const uint32_t start = (end > 5 ? end : 5) - 3;
    bool failed = false;
    for (uint32_t k = start; k <= end; k++) {
        double score;
        if (!run_k_dyn(k, m, n, D, coverage, rng, &score, lane)) {
            failed = true;
            break;
        }
        // result of this k: labels m.best, gains m.fbuf (unused downstream), used columns m.used
        if (k == 2) {
            // use_highest_gain (:673-693)
            double gbest = 0.0;
            uint32_t max_idx = 0;
            for (uint32_t d = 0; d < D; d++) {
                double gsum = 0.0;
                for (uint32_t i = 0; i < n; i++) gsum += jtk_fmax(m.data[i * D + d], 0.0);
                if (d == 0 || !(gsum < gbest)) {
                    gbest = gsum;
                    max_idx = d;
                }
            }
            for (uint32_t i = lane; i < n; i += 64) m.tmp_asn[i] = 0.0 < m.data[i * D + max_idx] ? 1 : 0;
            __syncthreads();
            // keep the mcmc result aside: fbuf is overwritten by get_read_lk_gains
            get_read_lk_gains<2>(m, n, D, m.tmp_asn, m.tmp_used, lane);
            double hscore = 0.0;
            for (uint32_t i = 0; i < n; i++) hscore += m.fbuf[i];
            if (score < hscore) {
                score = hscore;
                for (uint32_t i = lane; i < n; i += 64) m.best[i] = m.tmp_asn[i];
                for (uint32_t d = lane; d < D; d += 64) m.used[d] = m.tmp_used[d];
                __syncthreads();
            }
        }
        // expected_gains (:286-306)
        bool no_new = true;
        for (uint32_t d = 0; d < D; d++) no_new = no_new && (m.prev_used[d] == m.used[d]);
        double expt = 0.0;
        for (uint32_t d = 0; d < D; d++) {
            const bool check = ((!m.prev_used[d]) && m.used[d]) || no_new;
            const double v = check ? gains_expected(&params->gains, vt[2 * d], (int)vt[2 * d + 1]) : 0.0000001;
            if (d == 0 || !(v < expt)) expt = v;
        }
        const double expected_gain = jtk_fmax(0.8 * expt, 0.1) * per_cluster_cov + 0.1;
        if (expected_gain < score - max) {
            __syncthreads();
            for (uint32_t i = lane; i < n; i += 64) m.accepted[i] = m.best[i];
            for (uint32_t d = lane; d < D; d += 64) m.prev_used[d] = m.used[d];
            max = score;
            max_k = k;
            __syncthreads();
        } else {
            break;
        }
    }
    if (failed) {
        if (lane == 0) st->status = JTK_ERR_CHUNK_FAILED;
        return;
    }
    // ---- likelihood gains of the accepted clustering, re-assignment, posterior (:272, :98-105, :342-347)
    double *lg = lg_all + lg_off[ci];  // n x max_k
    likelihood_gain_dyn(max_k, m, n, D, m.accepted, lg, lane);
    __threadfence_block();
    for (uint32_t i = lane; i < n; i += 64) {
        double *lks = lg + (uint64_t)i * max_k;
        uint32_t asn = m.accepted[i], bi = 0;
        for (uint32_t c = 1; c < max_k; c++)
            if (!(lks[c] < lks[bi])) bi = c;
        if (lks[asn] + 0.001 < lks[bi]) asn = bi;
        // logsumexp (misc.rs:84-92)
        double mx = lks[0];
        for (uint32_t c = 1; c < max_k; c++)
            if (!(lks[c] < mx)) mx = lks[c];
        double sum = 0.0;
        for (uint32_t c = 0; c < max_k; c++) sum += jtk_exp(lks[c] - mx);
        const double total = mx + jtk_log(sum);
        label[i] = asn;
        for (uint32_t c = 0; c < post_stride; c++)
            post[(uint64_t)i * post_stride + c] = c < max_k ? lks[c] - total : 0.0;
    }
    if (lane == 0) {
        st->score = max;
        st->k = max_k;
    }
}

}  // namespace

size_t mcmc_lds_bytes(uint32_t lds_n, uint32_t lds_d) {
    auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
    size_t b = al((size_t)lds_n * lds_d * 8) + 2 * al((size_t)(lds_n + 1) * 8) + 2 * al((size_t)JTK_MAX_COPY * lds_d * 8) +
               2 * al((size_t)lds_n * 8) + al(lds_n + 1) + 5 * al(lds_n) + 3 * al(lds_d);
    return b;
}

void launch_mcmc(hipStream_t s, uint32_t n_chunks, const ChunkMeta *chunks, ChunkState *state,
                 const jtk_lc_params_t *params, const double *feat, const uint32_t *vtype, const uint64_t *vt_off,
                 uint32_t vt_stride_mode, uint32_t *label, double *post, uint32_t post_stride, double *lg,
                 const uint64_t *lg_off, uint32_t lds_n, uint32_t lds_d) {
    if (n_chunks == 0) return;
    const size_t lds = mcmc_lds_bytes(lds_n, lds_d);
    mcmc_kernel<<<n_chunks, 64, lds, s>>>(chunks, state, params, feat, vtype, vt_off, vt_stride_mode, label, post,
                                          post_stride, lg, lg_off, lds_n, lds_d);
}
