// gains.hip -- the stage preamble `estimate_gain` (likelihood_gains.rs:162-192, :253-315) with its 180,000 banded
// pair-HMM likelihoods on the device.
//
// For each of the 3 x homop_len (type, homopolymer length) profiles the reference simulates SAMPLE_NUM = 100
// pairs of sequences that differ by one variant, draws SEQ_NUM = 50 reads from each member of the pair and scores
// every read against both members with kiley's likelihood_antidiagonal_bootstrap(template, read, band):
//
//   host    the sampling (generate_seq, gen_diff_haplotypes, Generate::gen): sequential draws from one
//           Xoshiro256StarStar per simulation -- byte work, ~30 M draws in all;
//   device  edit_ops_kernel: the bootstrap alignment (global unit-cost DP + traceback), one wavefront per
//           (template, read) pair, the (len+1)^2 byte matrix in LDS;
//   device  band_prep + phmm_kernel's forward sweep (jtk_internal_likelihoods in session.hip): log P(read | template)
//           inside the band the alignment gives -- the same kernel the clustering path runs;
//   host    medians / percentiles of 100 numbers (likelihood_gains.rs:300-314).
//
// The pair-HMM is this build's own specification of the un-vendored kiley crate, and so are
// the bootstrap alignment's tie rules and the read sampler: parity of this entry point with the real kiley is
// unpinned; parity with oracle/likelihood_gains.c is exact.
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "device_common.h"

extern "C" int jtk_internal_likelihoods(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                                        const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                                        const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand,
                                        uint32_t radius, int device, double *lk_out);
extern "C" void jtk_internal_set_error(const char *msg);

namespace {

// ---- sampling: rand 0.8.5 on rand_xoshiro 0.6.0, as the reference's calls use them ---------------------------
struct Xoshiro {
    uint64_t s[4];
    explicit Xoshiro(uint64_t seed) {  // SeedableRng::seed_from_u64: four SplitMix64 outputs
        for (int i = 0; i < 4; i++) {
            seed += 0x9e3779b97f4a7c15ULL;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
            s[i] = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next_u64() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    // rand::seq::gen_index for a bound that fits u32: UniformInt<u32> single sampling on the upper half of a draw
    uint32_t gen_index(uint32_t n) {
        const uint32_t zone = (n << __builtin_clz(n)) - 1;
        for (;;) {
            const uint64_t m = (uint64_t)(uint32_t)(next_u64() >> 32) * n;
            if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
        }
    }
    // SliceRandom::choose_weighted: WeightedIndex<f64> over the running totals, one UniformFloat draw in [0, total)
    int choose_weighted(const double *w, int n) {
        double cum[4], total = w[0];
        for (int i = 1; i < n; i++) {
            cum[i - 1] = total;
            total += w[i];
        }
        if (!(total > 0.0)) return -1;
        double scale = total;
        const double max_rand = 1.0 - 0x1p-52;
        while (scale * max_rand + 0.0 >= total) {  // UniformFloat::new: shave ulps until the top draw stays below `high`
            uint64_t b;
            memcpy(&b, &scale, 8);
            b -= 1;
            memcpy(&scale, &b, 8);
        }
        const uint64_t bits = (next_u64() >> 12) | 0x3ff0000000000000ULL;
        double v;
        memcpy(&v, &bits, 8);
        const double chosen = (v - 1.0) * scale + 0.0;
        int lo = 0;
        while (lo < n - 1 && cum[lo] <= chosen) lo++;  // first index whose running total exceeds the draw
        return lo;
    }
};

const char BASES[] = "ACGT";
inline int code_of(uint8_t b) { return b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : 3; }

void generate_seq(Xoshiro &rng, size_t len, std::string &out) {  // kiley gen_seq::generate_seq
    for (size_t i = 0; i < len; i++) out.push_back(BASES[rng.gen_index(4)]);
}

// likelihood_gains.rs:213-251: two flanked homopolymer runs that differ by one substitution / deletion / insertion
void gen_diff_haplotypes(Xoshiro &rng, size_t len, int diff_type, std::string &hap1, std::string &hap2) {
    double w[4];
    const char homop = BASES[rng.gen_index(4)];
    for (int i = 0; i < 4; i++) w[i] = BASES[i] != homop ? 1.0 : 0.0;
    const char right = BASES[rng.choose_weighted(w, 4)];
    for (int i = 0; i < 4; i++) w[i] = (BASES[i] != homop && BASES[i] != right) ? 1.0 : 0.0;
    const char left = BASES[rng.choose_weighted(w, 4)];
    std::string c2(len, homop);
    for (int i = 0; i < 4; i++) w[i] = BASES[i] != homop ? 1.0 : 0.0;
    if (diff_type == JTK_DIFF_SUBST)
        c2[0] = BASES[rng.choose_weighted(w, 4)];
    else if (diff_type == JTK_DIFF_DEL)
        c2.erase(0, 1);
    else
        c2.insert(std::min<size_t>(1, len), 1, BASES[rng.choose_weighted(w, 4)]);
    hap1 = std::string(1, right) + std::string(len, homop) + std::string(1, left);
    hap2 = std::string(1, right) + c2 + std::string(1, left);
}

// Generate::gen of this build's pair-HMM specification: from Match at template position 0, draw the next state from
// the current state's transition row; Match emits and advances, Ins emits, Del advances; stop at the template's end.
void phmm_gen(const jtk_hmm_t &h, const std::string &tmpl, Xoshiro &rng, size_t cap, std::string &out) {
    size_t i = 0;
    int state = 0, prev = 4;
    out.clear();
    while (i < tmpl.size() && out.size() + 1 < cap) {
        const double tr[3][3] = {{h.mat_mat, h.mat_ins, h.mat_del}, {h.ins_mat, h.ins_ins, h.ins_del}, {h.del_mat, h.del_ins, h.del_del}};
        const int ns = rng.choose_weighted(tr[state], 3);
        if (ns < 0) break;
        state = ns;
        if (state == 2) {
            i++;
            continue;
        }
        const int b = rng.choose_weighted(state == 0 ? h.mat_emit + 4 * code_of((uint8_t)tmpl[i]) : h.ins_emit + 4 * prev, 4);
        if (b < 0) break;
        out.push_back(BASES[b]);
        prev = b;
        if (state == 0) i++;
    }
}

// ---- device: bootstrap alignment ---------------------------------------------------------------------------------
struct PairMeta {
    uint32_t tmpl_off, tmpl_len, read_off, read_len;
};
#define EDIT_MAX_LEN 250  // distances fit a byte
#define EDIT_OPS_STRIDE 512

// Global unit-cost alignment of one (template, read) pair per wavefront; ties resolve diagonal, then Del, then Ins
// on the traceback from the end (the stand-in for kiley's bootstrap alignment, oracle/phmm.c jo_edit_ops).
// LDS: the (tl+1) x (rl+1) distance matrix in bytes, filled one anti-diagonal per step.
__global__ __launch_bounds__(64) void edit_ops_kernel(const PairMeta *pairs, uint32_t n_pairs, const uint8_t *tmpl_all,
                                                      const uint8_t *read_all, uint8_t *ops_all, uint32_t *ops_len,
                                                      uint32_t max_w) {
    extern __shared__ uint8_t dm[];
    const uint32_t p = blockIdx.x, lane = threadIdx.x;
    if (p >= n_pairs) return;
    const PairMeta pm = pairs[p];
    const uint32_t tl = pm.tmpl_len, rl = pm.read_len, W = max_w;
    const uint8_t *x = tmpl_all + pm.tmpl_off, *y = read_all + pm.read_off;
    for (uint32_t j = lane; j <= rl; j += 64) dm[j] = (uint8_t)j;
    for (uint32_t i = lane; i <= tl; i += 64) dm[i * W] = (uint8_t)i;
    __syncthreads();
    for (uint32_t d = 2; d <= tl + rl; d++) {  // cells (i, d - i), 1 <= i <= tl, 1 <= d - i <= rl
        const uint32_t lo = d > rl ? d - rl : 1, hi = d - 1 < tl ? d - 1 : tl;
        for (uint32_t i = lo + lane; i <= hi; i += 64) {
            const uint32_t j = d - i;
            const uint32_t a = dm[(i - 1) * W + j - 1] + (x[i - 1] != y[j - 1] ? 1u : 0u);
            const uint32_t b = dm[(i - 1) * W + j] + 1u, c = dm[i * W + j - 1] + 1u;
            const uint32_t m = a < b ? a : b;
            dm[i * W + j] = (uint8_t)(m < c ? m : c);
        }
        __syncthreads();
    }
    if (lane == 0) {
        uint8_t *ops = ops_all + (size_t)p * EDIT_OPS_STRIDE;
        uint32_t i = tl, j = rl, k = 0;
        while (i > 0 || j > 0) {  // written back to front, reversed below
            const uint32_t here = dm[i * W + j];
            if (i > 0 && j > 0 && here == dm[(i - 1) * W + j - 1] + (x[i - 1] != y[j - 1] ? 1u : 0u)) {
                ops[k++] = x[i - 1] == y[j - 1] ? JTK_OP_MATCH : JTK_OP_MISMATCH;
                i--;
                j--;
            } else if (i > 0 && here == dm[(i - 1) * W + j] + 1u) {
                ops[k++] = JTK_OP_DEL;
                i--;
            } else {
                ops[k++] = JTK_OP_INS;
                j--;
            }
        }
        for (uint32_t a = 0; a < k / 2; a++) {
            const uint8_t t = ops[a];
            ops[a] = ops[k - 1 - a];
            ops[k - 1 - a] = t;
        }
        ops_len[p] = k;
    }
}

#define HIP_OK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            jtk_internal_set_error((std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
            return e_ == hipErrorOutOfMemory ? JTK_ERR_ALLOC : JTK_ERR_NO_DEVICE;             \
        }                                                                                     \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

// ops of every (template, read) pair, EDIT_OPS_STRIDE bytes apart
int device_edit_ops(const std::vector<PairMeta> &pairs, const std::vector<uint8_t> &tmpl, const std::vector<uint8_t> &reads,
                    uint32_t max_len, std::vector<uint8_t> &ops, std::vector<uint32_t> &ops_len) {
    const uint32_t n = (uint32_t)pairs.size(), W = max_len + 1;
    DevBuf d_pairs, d_tmpl, d_reads, d_ops, d_len;
    HIP_OK(hipMalloc(&d_pairs.p, pairs.size() * sizeof(PairMeta)));
    HIP_OK(hipMalloc(&d_tmpl.p, tmpl.size()));
    HIP_OK(hipMalloc(&d_reads.p, reads.size()));
    HIP_OK(hipMalloc(&d_ops.p, (size_t)n * EDIT_OPS_STRIDE));
    HIP_OK(hipMalloc(&d_len.p, (size_t)n * 4));
    HIP_OK(hipMemcpy(d_pairs.p, pairs.data(), pairs.size() * sizeof(PairMeta), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_tmpl.p, tmpl.data(), tmpl.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_reads.p, reads.data(), reads.size(), hipMemcpyHostToDevice));
    edit_ops_kernel<<<n, 64, (size_t)W * W, 0>>>((const PairMeta *)d_pairs.p, n, (const uint8_t *)d_tmpl.p,
                                                 (const uint8_t *)d_reads.p, (uint8_t *)d_ops.p, (uint32_t *)d_len.p, W);
    HIP_OK(hipGetLastError());
    ops.resize((size_t)n * EDIT_OPS_STRIDE);
    ops_len.resize(n);
    HIP_OK(hipMemcpy(ops.data(), d_ops.p, ops.size(), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ops_len.data(), d_len.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

double nth(std::vector<double> xs, size_t k) {  // select_nth_unstable_by(k).1: the k-th smallest
    std::sort(xs.begin(), xs.end());
    return xs[k];
}

// gain_of (likelihood_gains.rs:253-315) for a LIST of (length, type) profiles.  The reference calls gain_of once per profile
// (estimate_gain :170-181: 3 types x max homopolymer length, each 100 simulations x 2 haplotypes x 100 reads); the profiles are
// independent -- every simulation seeds its own generator (:269) -- so since round 6 they are simulated side by side on host
// threads, aligned by ONE edit_ops launch and scored by ONE likelihood session (rounds 2-5: nine sequential round trips of
// ~13 ms each through the device for 20,000 reads of 100 bases at a time; 119 of the stage's 371 ms of preambles).
struct GainJob {
    uint32_t len;
    int diff_type;
    jtk_gain_profile_t *out;
    // the simulations.  Per simulation i: templates [tmpl, diff]; reads 0..49 drawn from diff, 50..99 from tmpl
    std::vector<std::string> tmpls, reads;
    uint32_t max_len = 0;
};
const int GAIN_SAMPLE_NUM = 100, GAIN_SEQ_NUM = 50;  // likelihood_gains.rs:261-264

void simulate_gain_job(const jtk_lc_params_t &params, uint64_t seed, uint32_t seq_len, GainJob &job) {
    const int SAMPLE_NUM = GAIN_SAMPLE_NUM, SEQ_NUM = GAIN_SEQ_NUM;
    const size_t half = seq_len / 2, cap = 3 * (2 * half + 64) + 16;
    job.tmpls.assign(2 * SAMPLE_NUM, std::string());
    job.reads.assign((size_t)2 * SEQ_NUM * SAMPLE_NUM, std::string());
    uint32_t max_len = 0;
    for (int i = 0; i < SAMPLE_NUM; i++) {
        Xoshiro rng((uint64_t)i + seed);  // likelihood_gains.rs:269
        std::string seg1, seg2, h1, h2;
        generate_seq(rng, half, seg1);
        generate_seq(rng, half, seg2);
        gen_diff_haplotypes(rng, job.len, job.diff_type, h1, h2);
        job.tmpls[2 * i] = seg1 + h1 + seg2;
        job.tmpls[2 * i + 1] = seg1 + h2 + seg2;
        for (int t = 0; t < 2 * SEQ_NUM; t++) {
            const jtk_hmm_t &h = (t % 2 == 0) ? params.forward : params.reverse;  // :272-275 (SEQ_NUM is even)
            phmm_gen(h, job.tmpls[2 * i + (t < SEQ_NUM ? 1 : 0)], rng, cap, job.reads[(size_t)i * 2 * SEQ_NUM + t]);
            max_len = std::max<uint32_t>(max_len, (uint32_t)job.reads[(size_t)i * 2 * SEQ_NUM + t].size());
        }
        max_len = std::max<uint32_t>(max_len, (uint32_t)std::max(job.tmpls[2 * i].size(), job.tmpls[2 * i + 1].size()));
    }
    job.max_len = max_len;
}

int gains_of(const jtk_lc_params_t &params, uint64_t seed, uint32_t seq_len, uint32_t band, std::vector<GainJob> &jobs, int device) {
    const int SAMPLE_NUM = GAIN_SAMPLE_NUM, SEQ_NUM = GAIN_SEQ_NUM;
    // ---- host: the simulations, one thread per profile
    {
        std::vector<std::thread> th;
        for (size_t q = 1; q < jobs.size(); q++) th.emplace_back([&, q]() { simulate_gain_job(params, seed, seq_len, jobs[q]); });
        if (!jobs.empty()) simulate_gain_job(params, seed, seq_len, jobs[0]);
        for (auto &t : th) t.join();
    }
    uint32_t max_len = 0;
    for (const GainJob &job : jobs) max_len = std::max(max_len, job.max_len);
    if (max_len > EDIT_MAX_LEN) {
        jtk_internal_set_error("estimate_gain: simulated sequence longer than 250 bases");
        return JTK_ERR_UNSUPPORTED;
    }
    // ---- the batch: per profile, chunk 2i = (tmpl_i, its 100 reads), chunk 2i+1 = (diff_i, the same reads)
    const size_t chunks_per_job = (size_t)2 * SAMPLE_NUM, per = (size_t)2 * SEQ_NUM, n_chunks = chunks_per_job * jobs.size(),
                 n_reads = n_chunks * per;
    std::vector<jtk_lc_chunk_t> chunks(n_chunks);
    std::vector<uint8_t> tb, rb, strand(n_reads);
    std::vector<uint64_t> roff(1, 0);
    std::vector<PairMeta> pairs(n_reads);
    roff.reserve(n_reads + 1);
    for (size_t c = 0; c < n_chunks; c++) {
        const GainJob &job = jobs[c / chunks_per_job];
        const size_t cj = c % chunks_per_job;
        memset(&chunks[c], 0, sizeof chunks[c]);
        chunks[c].chunk_id = c;
        chunks[c].copy_num = 2;
        chunks[c].n_reads = (uint32_t)per;
        chunks[c].tmpl_off = tb.size();
        chunks[c].tmpl_len = job.tmpls[cj].size();
        chunks[c].read_first = c * per;
        tb.insert(tb.end(), job.tmpls[cj].begin(), job.tmpls[cj].end());
        for (size_t t = 0; t < per; t++) {
            const std::string &r = job.reads[(cj / 2) * per + t];
            PairMeta &pm = pairs[c * per + t];
            pm.tmpl_off = (uint32_t)chunks[c].tmpl_off;
            pm.tmpl_len = (uint32_t)job.tmpls[cj].size();
            pm.read_off = (uint32_t)rb.size();
            pm.read_len = (uint32_t)r.size();
            rb.insert(rb.end(), r.begin(), r.end());
            roff.push_back(rb.size());
            strand[c * per + t] = (t % 2 == 0) ? 1 : 0;
        }
    }
    // ---- device: bootstrap alignments, then the banded likelihoods
    std::vector<uint8_t> ops_strided;
    std::vector<uint32_t> ops_len;
    int rc = device_edit_ops(pairs, tb, rb, max_len, ops_strided, ops_len);
    if (rc) return rc;
    std::vector<uint8_t> ops;
    std::vector<uint64_t> ooff(1, 0);
    ooff.reserve(n_reads + 1);
    {
        size_t total = 0;
        for (size_t g = 0; g < n_reads; g++) total += ops_len[g];
        ops.reserve(total);
    }
    for (size_t g = 0; g < n_reads; g++) {
        ops.insert(ops.end(), ops_strided.begin() + g * EDIT_OPS_STRIDE, ops_strided.begin() + g * EDIT_OPS_STRIDE + ops_len[g]);
        ooff.push_back(ops.size());
    }
    std::vector<double> lk(n_reads);
    rc = jtk_internal_likelihoods(&params, n_chunks, chunks.data(), tb.data(), rb.data(), roff.data(), ops.data(), ooff.data(),
                                  strand.data(), band, device, lk.data());
    if (rc) return rc;
    // ---- host: likelihood_gains.rs:276-314, per profile
    for (size_t q = 0; q < jobs.size(); q++) {
        const double *lkq = lk.data() + q * chunks_per_job * per;
        std::vector<double> medians(SAMPLE_NUM), probs(SAMPLE_NUM);
        for (int i = 0; i < SAMPLE_NUM; i++) {
            const double *base = &lkq[(size_t)(2 * i) * per], *dif = &lkq[(size_t)(2 * i + 1) * per];
            std::vector<double> lk_diff(SEQ_NUM);
            for (int t = 0; t < SEQ_NUM; t++) lk_diff[t] = dif[t] - base[t];
            const double expected_gain = nth(lk_diff, SEQ_NUM / 2);
            const double min_gain = jobs[q].diff_type == JTK_DIFF_SUBST ? expected_gain / 10.0 : 0.0001;
            int null_cnt = 0;
            for (int t = SEQ_NUM; t < 2 * SEQ_NUM; t++) null_cnt += (base[t] + min_gain < dif[t]) ? 1 : 0;
            medians[i] = expected_gain;
            probs[i] = (double)null_cnt / (double)SEQ_NUM;
        }
        jobs[q].out->gain = nth(medians, SAMPLE_NUM / 10);
        jobs[q].out->prob = std::max(nth(probs, SAMPLE_NUM * 2 / 3), 0.000000001);
    }
    return 0;
}

// kiley gen_seq::introduce_errors(seq, rng, 0, 1, 0) (likelihood_gains.rs:18), own specification (oracle/likelihood_gains.c
// introduce_one_deletion): shuffle [Match x (len - 1), Del] with SliceRandom::shuffle, drop the base at the Del's place
void introduce_one_deletion(Xoshiro &rng, const std::string &seq, std::string &out) {
    const size_t len = seq.size();
    size_t del_pos = len - 1;
    for (size_t i = len - 1; i >= 1; i--) {
        const size_t j = rng.gen_index((uint32_t)(i + 1));
        if (del_pos == i)
            del_pos = j;
        else if (del_pos == j)
            del_pos = i;
    }
    out = seq;
    out.erase(del_pos, 1);
}

// estimate_minimum_gain's samples [s0, s1): medians[s] = median over reads of lk(read | hap1) - lk(read | hap2)
int minimum_gain_batch(const jtk_lc_params_t &params, uint64_t seed, size_t s0, size_t s1, uint32_t seq_num, uint32_t len,
                       uint32_t band, int device, double *medians) {
    const size_t ns = s1 - s0, cap = 3 * (size_t)len + 16;
    std::vector<std::string> tmpls(2 * ns), reads(ns * seq_num);
    uint32_t max_len = len;
    for (size_t i = 0; i < ns; i++) {
        Xoshiro rng(seed + (uint64_t)(s0 + i));  // likelihood_gains.rs:16
        generate_seq(rng, len, tmpls[2 * i]);
        introduce_one_deletion(rng, tmpls[2 * i], tmpls[2 * i + 1]);
        for (uint32_t t = 0; t < seq_num; t++) {
            const jtk_hmm_t &h = (t % 2 == 0) ? params.forward : params.reverse;  // :22-25
            phmm_gen(h, tmpls[2 * i], rng, cap, reads[i * seq_num + t]);
            max_len = std::max<uint32_t>(max_len, (uint32_t)reads[i * seq_num + t].size());
        }
    }
    if (max_len > EDIT_MAX_LEN) {
        jtk_internal_set_error("estimate_minimum_gain: simulated sequence longer than 250 bases");
        return JTK_ERR_UNSUPPORTED;
    }
    const size_t n_chunks = 2 * ns, n_reads = n_chunks * seq_num;
    std::vector<jtk_lc_chunk_t> chunks(n_chunks);
    std::vector<uint8_t> tb, rb, strand(n_reads);
    std::vector<uint64_t> roff(1, 0);
    std::vector<PairMeta> pairs(n_reads);
    for (size_t c = 0; c < n_chunks; c++) {
        memset(&chunks[c], 0, sizeof chunks[c]);
        chunks[c].chunk_id = c;
        chunks[c].copy_num = 2;
        chunks[c].n_reads = seq_num;
        chunks[c].tmpl_off = tb.size();
        chunks[c].tmpl_len = tmpls[c].size();
        chunks[c].read_first = c * seq_num;
        tb.insert(tb.end(), tmpls[c].begin(), tmpls[c].end());
        for (uint32_t t = 0; t < seq_num; t++) {
            const std::string &r = reads[(c / 2) * seq_num + t];
            PairMeta &pm = pairs[c * seq_num + t];
            pm.tmpl_off = (uint32_t)chunks[c].tmpl_off;
            pm.tmpl_len = (uint32_t)tmpls[c].size();
            pm.read_off = (uint32_t)rb.size();
            pm.read_len = (uint32_t)r.size();
            rb.insert(rb.end(), r.begin(), r.end());
            roff.push_back(rb.size());
            strand[c * seq_num + t] = (t % 2 == 0) ? 1 : 0;
        }
    }
    std::vector<uint8_t> ops_strided;
    std::vector<uint32_t> ops_len;
    int rc = device_edit_ops(pairs, tb, rb, max_len, ops_strided, ops_len);
    if (rc) return rc;
    std::vector<uint8_t> ops;
    std::vector<uint64_t> ooff(1, 0);
    for (size_t g = 0; g < n_reads; g++) {
        ops.insert(ops.end(), ops_strided.begin() + g * EDIT_OPS_STRIDE, ops_strided.begin() + g * EDIT_OPS_STRIDE + ops_len[g]);
        ooff.push_back(ops.size());
    }
    std::vector<double> lk(n_reads);
    rc = jtk_internal_likelihoods(&params, n_chunks, chunks.data(), tb.data(), rb.data(), roff.data(), ops.data(), ooff.data(),
                                  strand.data(), band, device, lk.data());
    if (rc) return rc;
    for (size_t i = 0; i < ns; i++) {
        std::vector<double> d(seq_num);
        for (uint32_t t = 0; t < seq_num; t++) d[t] = lk[(2 * i) * seq_num + t] - lk[(2 * i + 1) * seq_num + t];
        medians[i] = nth(d, seq_num / 2);
    }
    return 0;
}

}  // namespace

extern "C" {

// estimate_minimum_gain (likelihood_gains.rs:6-39): what correct_clustering's protection rule is scaled by
// (phmm_likelihood_correction.rs:118).  The reference's constants are (23908, 1000, 500, 100, 25).
int jtk_lc_estimate_minimum_gain(const jtk_hmm_t *forward, const jtk_hmm_t *reverse, uint64_t seed, uint32_t sample_num,
                                 uint32_t seq_num, uint32_t len, uint32_t band, double *out, int device) {
    jtk_internal_set_error("");
    if (!forward || !reverse || !out || sample_num < 3 || seq_num == 0 || len < 2 || len > 200 || band == 0 ||
        band > JTK_MAX_RADIUS) {
        jtk_internal_set_error("jtk_lc_estimate_minimum_gain: bad argument (sample_num >= 3, 2 <= len <= 200, 1 <= band <= 30)");
        return JTK_ERR_INVALID_ARG;
    }
    if (!jtk_lc_device_ok(device)) {
        jtk_internal_set_error("no gfx950 device (jtk_lc has no CPU fallback)");
        return JTK_ERR_NO_DEVICE;
    }
    HIP_OK(hipSetDevice(device));
    jtk_lc_params_t params;
    memset(&params, 0, sizeof params);
    params.forward = *forward;
    params.reverse = *reverse;
    params.gains.max_homopolymer_len = 1;  // unused by the likelihood batches
    params.haploid_coverage = 1.0;
    params.band_frac = 0.0;
    std::vector<double> medians(sample_num);
    // batches of ~100,000 reads keep the likelihood session's workspaces at a few GB
    const size_t per = std::max<size_t>(1, 50000 / seq_num);
    for (size_t s0 = 0; s0 < sample_num; s0 += per) {
        const size_t s1 = std::min<size_t>(sample_num, s0 + per);
        const int rc = minimum_gain_batch(params, seed, s0, s1, seq_num, len, band, device, medians.data() + s0);
        if (rc) return rc;
    }
    std::sort(medians.begin(), medians.end());
    *out = std::max(medians[2], 1.0);
    return 0;
}

int jtk_lc_estimate_gains(const jtk_hmm_t *forward, const jtk_hmm_t *reverse, uint64_t seed, uint32_t seq_len,
                          uint32_t band, uint32_t homop_len, jtk_gains_t *out, int device) {
    jtk_internal_set_error("");
    if (!forward || !reverse || !out || homop_len == 0 || homop_len > JTK_GAINS_MAX_HOMOP || seq_len < 2 || band == 0 ||
        band > JTK_MAX_RADIUS) {
        jtk_internal_set_error("jtk_lc_estimate_gains: bad argument (1 <= homop_len <= 8, 1 <= band <= 30)");
        return JTK_ERR_INVALID_ARG;
    }
    if (!jtk_lc_device_ok(device)) {
        jtk_internal_set_error("no gfx950 device (jtk_lc has no CPU fallback)");
        return JTK_ERR_NO_DEVICE;
    }
    HIP_OK(hipSetDevice(device));
    jtk_lc_params_t params;
    memset(&params, 0, sizeof params);
    params.forward = *forward;
    params.reverse = *reverse;
    params.gains.max_homopolymer_len = 1;  // unused by the likelihood batches
    params.haploid_coverage = 1.0;
    params.band_frac = 0.0;
    memset(out, 0, sizeof *out);
    out->max_homopolymer_len = homop_len;
    const int types[3] = {JTK_DIFF_SUBST, JTK_DIFF_DEL, JTK_DIFF_INS};  // likelihood_gains.rs:170-181
    jtk_gain_profile_t *dst[3] = {out->subst, out->deletions, out->insertions};
    std::vector<GainJob> jobs;
    for (int ty = 0; ty < 3; ty++)
        for (uint32_t len = 1; len <= homop_len; len++) {
            GainJob j;
            j.len = len;
            j.diff_type = types[ty];
            j.out = &dst[ty][len - 1];
            jobs.push_back(std::move(j));
        }
    {
        const int rc = gains_of(params, seed, seq_len, band, jobs, device);
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
