// session.hip -- resident-batch session and the C-ABI entry points of include/jtk_lc.h.
//
// One process drives one GPU.  A session validates and encodes the flat host batch, uploads it once, allocates
// every workspace up front (sized for 288 GB of HBM: the full N x 14(L+1) tables of all chunks stay
// resident), and then runs the stage as a short sequence of kernel launches on its own stream:
//
//   band_prep -> [ phmm -> finalize -> polish_round ]* until no chunk is active -> filter -> mcmc
//
// (mod.rs:86-123 per chunk).  The last polishing pass that finds no edit has produced exactly the
// modification table `clustering` would recompute (same consensus, same ops, same radius: mod.rs:105 vs
// :112, pseudo_mcmc.rs:117), so it is reused instead of recomputed.
//
// Chunks with copy_num >= 8 take clustering_recursive's split branch (mod.rs:138-189): the batch pass clusters
// them into at most four groups, and run_split() below then drives the sub-problems -- polish the group's
// consensus, cluster it with its share of the copies, recurse -- as further resident batches, one sub-problem
// per chunk and round because a chunk's calls share one RNG stream in depth-first order.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <functional>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <algorithm>
#include <atomic>
#include <vector>

#include "device_common.h"
#include "jtk_lc_debug.h"

#define JTK_POOL_DEVICES 16

// launchers defined in the other translation units
void launch_filter(hipStream_t s, uint32_t n_chunks, const ReadMeta *reads, const ChunkMeta *chunks,
                   ChunkState *state, DevBufs bufs, const jtk_lc_params_t *params, const double *table,
                   uint16_t *homop, const uint64_t *homop_off, double *aux, const uint64_t *aux_off, double *cand,
                   uint32_t *list, uint8_t *sel, double *feat, uint32_t *vtype, uint32_t *pos, uint32_t max_tmpl,
                   const HmmDev *hmm2, const int *rawG, const double *lk, int fused);
void launch_pick_trace(hipStream_t s, uint32_t ci, const ReadMeta *reads, const ChunkMeta *chunks, ChunkState *state,
                       const jtk_lc_params_t *params, const double *table, const uint16_t *homop, const uint64_t *homop_off,
                       const double *cand, uint32_t *list, uint8_t *sel, double *feat, uint32_t *vtype, uint32_t *pos,
                       const HmmDev *hmm2, const int *rawG, const double *lk, int fused, uint32_t *tr, uint32_t *tr_count);
size_t mcmc_trace_doubles(uint32_t n_reads);
int launch_mcmc_trace(hipStream_t s, const ChunkMeta *chunks, ChunkState *state, const jtk_lc_params_t *params, const double *feat,
                      const uint32_t *vtype, uint32_t *label, double *post, uint32_t post_stride, double *lg, const uint64_t *lg_off,
                      uint32_t n, uint32_t d, uint32_t k, const uint32_t *order, unsigned char *ws, const uint64_t *ws_off,
                      double *trace);
size_t mcmc_lds_bytes(uint32_t lds_n, uint32_t lds_d, uint32_t lds_k);
size_t mcmc_ws_bytes(uint32_t n, uint32_t d, uint32_t k);
int launch_mcmc_huge(hipStream_t s, uint32_t n_chunks, const ChunkMeta *chunks, ChunkState *state, const jtk_lc_params_t *params,
                     const double *feat, const uint32_t *vtype, const uint64_t *vt_off, uint32_t vt_stride_mode, uint32_t *label,
                     double *post, uint32_t post_stride, double *lg, const uint64_t *lg_off, uint32_t max_n, uint32_t max_d,
                     uint32_t max_k, const uint64_t *rng_resume, const uint32_t *order, unsigned char *ws, const uint64_t *ws_off);
int launch_mcmc(hipStream_t s, uint32_t n_chunks, const ChunkMeta *chunks, ChunkState *state,
                const jtk_lc_params_t *params, const double *feat, const uint32_t *vtype, const uint64_t *vt_off,
                uint32_t vt_stride_mode, uint32_t *label, double *post, uint32_t post_stride, double *lg,
                const uint64_t *lg_off, uint32_t lds_n, uint32_t lds_d, uint32_t lds_k, const uint64_t *rng_resume,
                const uint32_t *order, uint32_t *split, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join);

namespace {

thread_local std::string g_last_error;
thread_local jtk_lc_timing_t g_timing;

int fail(int status, const std::string &msg) {
    g_last_error = msg;
    return status;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(_e == hipErrorOutOfMemory ? JTK_ERR_ALLOC : JTK_ERR_NO_DEVICE,         \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                    \
    } while (0)

inline int base_code(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

HmmDev to_dev(const jtk_hmm_t &h) {
    HmmDev d;
    const double a[9] = {h.mat_mat, h.mat_ins, h.mat_del, h.ins_mat, h.ins_ins, h.ins_del, h.del_mat, h.del_ins, h.del_del};
    memcpy(d.a, a, sizeof a);
    memcpy(d.eM, h.mat_emit, sizeof d.eM);
    memcpy(d.eI, h.ins_emit, sizeof d.eI);
    return d;
}

// Device blocks of destroyed sessions are kept for the next session on the same device: a stage call is one-shot
// (create, run, fetch, destroy) and is entered several times per pipeline with batches of similar shape, and mapping
// the ~88 GB of workspaces of 2500 chunks costs ~0.2 s on a fresh device and ~2.4 s once the same memory has been
// freed before (the driver scrubs it) -- more than the kernels take.  Nothing relies on the contents of a fresh
// block.  jtk_lc_trim_cache() returns everything to the driver.
struct BlockPool {
    // Small blocks are kept too (round 6): a session holds ~40 of them (metadata, counters, offsets), hipFree of each one
    // synchronises EVERY stream of the device, and the model refit creates ten sessions in a row.  Below 1 MiB a request is
    // rounded up to a power of two (dev_alloc), so the few size classes match exactly.
    static const size_t MIN_BYTES = 256;
    static const size_t SMALL_BYTES = 1u << 20;
    // per device: JTK_LC_POOL_GB (default 32: the pool must not starve torch / RCCL / another library in the same process;
    // blocks beyond the cap go straight back to the driver), 0 disables pooling
    static size_t max_cached() {
        static const size_t v = []() -> size_t {
            const char *e = getenv("JTK_LC_POOL_GB");
            const double gb = e ? atof(e) : 32.0;
            return gb > 0.0 ? (size_t)(gb * 1073741824.0) : 0;
        }();
        return v;
    }
    std::mutex m;
    std::multimap<size_t, void *> blocks[JTK_POOL_DEVICES];
    size_t cached[JTK_POOL_DEVICES] = {};
    void *take(int dev, size_t bytes, size_t *cap) {
        std::lock_guard<std::mutex> lock(m);
        auto &b = blocks[dev];
        auto it = b.lower_bound(bytes);
        if (it == b.end() || it->first > bytes + bytes / 4 + (bytes < SMALL_BYTES ? 0 : SMALL_BYTES)) return nullptr;
        void *p = it->second;
        *cap = it->first;
        cached[dev] -= it->first;
        b.erase(it);
        return p;
    }
    bool give(int dev, void *p, size_t cap) {
        std::lock_guard<std::mutex> lock(m);
        if (cap < MIN_BYTES || cached[dev] + cap > max_cached()) return false;
        blocks[dev].emplace(cap, p);
        cached[dev] += cap;
        return true;
    }
    void trim(int dev) {
        std::lock_guard<std::mutex> lock(m);
        for (auto &kv : blocks[dev]) (void)hipFree(kv.second);
        blocks[dev].clear();
        cached[dev] = 0;
    }
};
BlockPool g_pool;

// A batch in flight uses up to four slices x two streams (the chain's general kernel runs beside its light one).  ROCm maps
// streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that share a queue run in order: a slice's 300 ms
// chain kernel would then hold up another slice's pair-HMM launches.  Eight streams need eight queues, and the host's own
// streams (the null stream as soon as anything uses it) take theirs: 12.  The variable is read when the HIP runtime
// initialises, so it is set -- unless the host has set it -- when this library is loaded; a host that has initialised HIP
// earlier sets it itself.
static bool g_queues_by_library = false;  // the variable was absent when the library was loaded
static int g_host_queues = 0;             // the host's own setting, if any
__attribute__((constructor)) void jtk_lc_default_hw_queues() {
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    if (q) {
        g_host_queues = atoi(q);
    } else {
        g_queues_by_library = true;
        setenv("GPU_MAX_HW_QUEUES", "12", 0);
    }
}

struct DevPtr {
    void *p = nullptr;
    size_t cap = 0;
    int dev = -1;
    ~DevPtr() {
        if (!p) return;
        if (dev < 0 || dev >= JTK_POOL_DEVICES || !g_pool.give(dev, p, cap)) (void)hipFree(p);
    }
    template <typename T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
};

// The forward scratch of phmm_kernel / phmm_pair_kernel (device_common.h: StripeSet): one set of stripes per device, shared by
// every session on it -- four slices in flight used to hold four sets of 3,072 x 4.1 MB, of which the device's resident waves
// could only ever use one set's worth.  A session that needs longer stripes than the current set has replaces it (sessions
// that still run keep theirs through the shared_ptr).
struct StripePool {
    DevPtr mem, owner;
    uint64_t stride = 0;  // doubles
    uint32_t n = 0;
    StripeSet set() const {
        static const uint32_t unfenced = getenv("JTK_STRIPE_UNFENCED") ? 1u : 0u;  // measurement only
        return StripeSet{mem.as<double>(), stride, owner.as<uint32_t>(), n, unfenced};
    }
};
std::mutex g_stripe_mutex;
std::shared_ptr<StripePool> g_stripes[JTK_POOL_DEVICES];

// The pair-HMM gate (round 6).  A pass of a batch is two phases with opposite needs: the polish rounds + tables + filter fill the
// device (or could), the chain kernels that follow keep a few hundred long-lived workgroups busy for 100-300 ms and leave the
// rest of the device idle.  Slices that start together STAY together: they share the device during their pair-HMM rounds, finish
// them at the same time and then all sit in their chain kernels at once (profiles/r06_trace_summary.txt: 200-300 ms per
// 880 ms step in which the device ran nothing but chain workgroups; without the chain kernels the same step takes 627 ms).
// The gate admits at most JTK_LC_PHASE_SLOTS batches per device to the pair-HMM phase at a time, first come first served; a
// batch leaves it when its chain kernels are queued.  The admitted batches get the whole device, finish their rounds sooner, and
// their chains run under the next batches' pair-HMM rounds.  MEASURED AND NOT ADOPTED (profiles/r06_gate.txt: 2,810 / 2,888 /
// 2,982 chunks/s with 2 / 3 / 4 slots against 2,902 without; start offsets between the slices, r06_stagger.txt, do nothing
// either): a chain workgroup that runs beside pair-HMM waves takes their registers and LDS for as long as it lives, which costs
// about what the idle tail of the lockstep costs.  Default 0 = no gate; the switch stays for measurements.  No effect on results.
struct PhaseGate {
    std::mutex m;
    std::condition_variable cv;
    uint64_t next_ticket = 0, released = 0;
};
PhaseGate g_gate[JTK_POOL_DEVICES];
int phase_slots() {
    static const int v = []() {
        const char *e = getenv("JTK_LC_PHASE_SLOTS");
        return e ? atoi(e) : 0;
    }();
    return v;
}
struct PhaseHold {
    PhaseGate *g = nullptr;
    explicit PhaseHold(int device) {
        if (phase_slots() <= 0 || device < 0 || device >= JTK_POOL_DEVICES) return;  // 0: no gate (rounds 1-5)
        g = &g_gate[device];
        std::unique_lock<std::mutex> lock(g->m);
        const uint64_t ticket = g->next_ticket++;
        g->cv.wait(lock, [&]() { return ticket < g->released + (uint64_t)phase_slots(); });
    }
    void release() {
        if (!g) return;
        {
            std::lock_guard<std::mutex> lock(g->m);
            g->released++;
        }
        g->cv.notify_all();
        g = nullptr;
    }
    ~PhaseHold() { release(); }
};

struct KernelTimer {
    hipEvent_t a = nullptr, b = nullptr;
    int kind = 0;
};

// result of one clustering_recursive call (ClusteringDevResult, mod.rs:124)
struct SplitResult {
    std::vector<uint32_t> asn;
    std::vector<double> post;  // n x k log-posteriors
    uint32_t k = 0;
    double score = 0.0;
    int status = 0;
};

// what a sub-problem inherits from its chunk instead of deriving it from its own template / read count
struct ChunkExtra {
    uint32_t radius;        // config.band_width (mod.rs:112,142,153)
    double local_coverage;  // config.local_coverage (mod.rs:108-112)
    uint64_t rng[4];        // the chunk's generator, as the previous call left it (mod.rs:158)
    uint32_t take_num;      // HMMPolishConfig take_num (0 = every read votes)
};

}  // namespace

struct ChainClass {
    uint32_t first = 0, count = 0, lds_n = 0, lds_d = 0, lds_k = 0, lds_bytes = 0;
};

// The mapped pinned pages the polish rounds report into (JTK_NACTIVE_SLOTS counters per session): taken from and returned to
// a process-wide free list -- sessions are created per slice per one-shot call, and hipHostFree synchronises the device,
// which stalls the other slices' streams.  The pages are freed with the process.
static std::mutex g_pinned_mutex;
static std::vector<uint32_t *> g_pinned_free;
static uint32_t *pinned_page_take() {
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        if (!g_pinned_free.empty()) {
            uint32_t *p = g_pinned_free.back();
            g_pinned_free.pop_back();
            return p;
        }
    }
    uint32_t *p = nullptr;
    if (hipHostMalloc(reinterpret_cast<void **>(&p), JTK_NACTIVE_SLOTS * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess)
        return nullptr;
    return p;
}
static void pinned_page_give(uint32_t *p) {
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    g_pinned_free.push_back(p);
}

struct jtk_lc_session {
    int device = 0;
    hipStream_t stream = nullptr;
    jtk_lc_params_t params;
    uint32_t n_chunks = 0, n_reads = 0, post_stride = 1;
    uint32_t max_tmpl = 0, max_read = 0, max_n = 0, max_copy = 0, n_waves = 0;
    ChainClass chain_class[3];  // the chain kernel's launches (by LDS need), as ranges of d_order; [2]: the pile-ups whose
                                // work area lives in global memory (mcmc_kernel_huge: more than JTK_MAX_PILEUP reads, or more
                                // than a CU's LDS)
    DevPtr d_chain_ws, d_chain_ws_off;
    uint32_t n_pair_items = 0, n_pair_waves = 0;  // phmm_pair_kernel: chunks with band radius <= JTK_PAIR_MAX_RADIUS
    DevPtr d_pair_items;
    std::shared_ptr<StripePool> stripes;  // the device's forward scratch (shared)
    bool features_only = false;
    std::vector<ChunkMeta> h_chunks;
    std::vector<ReadMeta> h_reads;
    std::vector<ChunkState> h_state0;  // initial state (re-uploaded at every run)
    std::vector<uint64_t> h_in_tmpl_off;
    // device memory
    DevPtr d_params, d_hmm2, d_chunks, d_reads, d_state, d_tmpl0, d_tmpl1, d_ops0, d_ops1, d_opslen0, d_opslen1,
        d_ey, d_delta, d_raw, d_rawG, d_lk, d_total, d_edits, d_newlen, d_counter, d_nactive,
        d_homop, d_homop_off, d_aux, d_aux_off, d_cand, d_list, d_sel, d_feat, d_vtype, d_pos, d_label, d_post,
        d_lg, d_lg_off, d_vt_off, d_tmpl_init, d_ops_init, d_opslen_init, d_order;
    size_t tmpl_bytes = 0, ops_bytes = 0;
    DevBufs bufs;
    std::vector<KernelTimer> timers;
    // clustering_recursive (mod.rs:125-189)
    uint32_t ignore_edge = 3;            // HMMPolishConfig ignore_edge: 3 for a chunk (mod.rs:105), 0 for a sub-problem (:153)
    bool has_split = false;              // some chunk has copy_num >= UPPER_COPY_NUM
    std::vector<uint32_t> h_copy0;       // Chunk.copy_num as given (ChunkMeta.copy_num is what one clustering() call sees)
    std::vector<uint8_t> h_read_bases, h_strand;  // kept on the host only when has_split
    std::vector<uint64_t> h_read_off;
    std::vector<SplitResult> split;      // per chunk; .k == 0: not a split chunk
    DevPtr d_rng;                        // 4 x u64 per chunk: where each chunk's RNG stream resumes (sub-problems only)
    bool resume_rng = false;
    bool polish_only = false;            // jtk_lc_polish_chunks: no variant search, no clustering
    bool ran = false, ran_fused = false; // a clustering pass has run (jtk_lc_session_trace needs its device state); with the fused filter?
    // reads whose band is wider than one wavefront (radius > JTK_MAX_RADIUS) take phmm_wide_kernel
    uint32_t n_wide_reads = 0, max_wide_radius = 0, n_wide_waves = 0;
    uint64_t wide_stride = 0;
    DevPtr d_wide_scratch, d_wide_counter;
    DevPtr d_state0;                     // pristine per-chunk state: a pass begins with a device-side copy of it
    // the variable-length outputs of a fetch, packed on the device (io_kernels.hip): lengths per read / chunk, their prefix sums,
    // the re-threaded ops and the consensus as the caller gets them; allocated by the first fetch that asks for them
    DevPtr d_out_len, d_out_off, d_out_ops, d_out_cons;
    // host mirrors of the never-reset device ticket counters of the work queues (device_common.h): d_counter[0] phmm_kernel,
    // d_counter[1] phmm_pair_kernel, d_wide_counter[0] phmm_wide_kernel
    uint32_t tk_phmm = 0, tk_pair = 0, tk_wide = 0;
    uint32_t *h_nactive = nullptr;       // pinned + mapped: the per-round "chunks still active" counters, written by commit_kernel
    uint32_t *h_nactive_dev = nullptr;   // the same memory as the device addresses it
    hipEvent_t ev_round[2] = {nullptr, nullptr};
    // the chain launch: light / general chunk lists made on the device (mcmc_kernels.hip), the general kernel on its own stream
    DevPtr d_chain_split;
    hipStream_t side = nullptr, chain_main = nullptr;   // chain_main: experiment JTK_CHAIN_CUS (CU-masked chain streams)
    hipEvent_t ev_chain[4] = {nullptr, nullptr, nullptr, nullptr};
    ~jtk_lc_session() {
        if (stream) (void)hipStreamSynchronize(stream);  // blocks go back to the pool, not through hipFree's implicit sync
        if (side) {
            (void)hipStreamSynchronize(side);
            (void)hipStreamDestroy(side);
        }
        if (chain_main) {
            (void)hipStreamSynchronize(chain_main);
            (void)hipStreamDestroy(chain_main);
        }
        for (auto &e : ev_chain)
            if (e) (void)hipEventDestroy(e);
        for (auto &t : timers) {
            if (t.a) (void)hipEventDestroy(t.a);
            if (t.b) (void)hipEventDestroy(t.b);
        }
        if (stream) (void)hipStreamDestroy(stream);
        if (h_nactive) pinned_page_give(h_nactive);  // (hipHostFree synchronises the whole device: never on the one-shot path)
        for (auto &e : ev_round)
            if (e) (void)hipEventDestroy(e);
    }
};

namespace {

template <typename T>
int dev_alloc(DevPtr &d, size_t count) {
    size_t bytes = (count ? count : 1) * sizeof(T);
    if (bytes < BlockPool::SMALL_BYTES) {   // size classes for the small blocks (see BlockPool)
        size_t cls = BlockPool::MIN_BYTES;
        while (cls < bytes) cls <<= 1;
        bytes = cls;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < JTK_POOL_DEVICES && bytes >= BlockPool::MIN_BYTES) {
        d.p = g_pool.take(dev, bytes, &d.cap);
        if (d.p) {
            d.dev = dev;
            return 0;
        }
    }
    hipError_t e = hipMalloc(&d.p, bytes);
    if (e == hipErrorOutOfMemory && dev >= 0 && dev < JTK_POOL_DEVICES) {  // cached blocks of another shape are in the way
        (void)hipGetLastError();
        g_pool.trim(dev);
        e = hipMalloc(&d.p, bytes);
    }
    HIP_TRY(e);
    d.cap = bytes;
    d.dev = dev;
    return 0;
}
template <typename T>
int dev_upload(jtk_lc_session *s, DevPtr &d, const std::vector<T> &v) {
    int rc = dev_alloc<T>(d, v.size());
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s->stream));
    return 0;
}

int pick_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(JTK_ERR_NO_DEVICE, "no HIP device visible (jtk_lc has no CPU fallback)");
    if (device < 0 || device >= count) return fail(JTK_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(JTK_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    return 0;
}

void tstart(jtk_lc_session *s, int kind) {
    KernelTimer t;
    t.kind = kind;
    (void)hipEventCreate(&t.a);
    (void)hipEventCreate(&t.b);
    (void)hipEventRecord(t.a, s->stream);
    s->timers.push_back(t);
}
void tstop(jtk_lc_session *s) { (void)hipEventRecord(s->timers.back().b, s->stream); }

}  // namespace

extern "C" {

static int session_create_ex(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                             const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                             const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand,
                             uint32_t post_stride, int device, const ChunkExtra *extra, uint32_t ignore_edge,
                             jtk_lc_session_t **out, bool polish_only = false) {
    g_last_error.clear();
    if (!params || !out || (n_chunks && (!chunks || !tmpl_bases || !read_bases || !read_off || !ops || !ops_off || !strand)))
        return fail(JTK_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (post_stride == 0) return fail(JTK_ERR_INVALID_ARG, "post_stride must be >= 1");
    if (params->gains.max_homopolymer_len == 0 || params->gains.max_homopolymer_len > JTK_GAINS_MAX_HOMOP)
        return fail(JTK_ERR_INVALID_ARG, "gains.max_homopolymer_len out of range");
    int rc = pick_device(device);
    if (rc) return rc;
    jtk_lc_session *s = new jtk_lc_session();
    std::unique_ptr<jtk_lc_session> guard(s);
    s->device = device;
    s->params = *params;
    s->post_stride = post_stride;
    s->n_chunks = (uint32_t)n_chunks;
    s->ignore_edge = ignore_edge;
    s->polish_only = polish_only;  // no variant search, no chain: none of their limits or workspaces apply
    s->h_copy0.resize(n_chunks);
    HIP_TRY(hipStreamCreate(&s->stream));

    // ---- host-side layout + validation + encoding
    uint64_t n_reads = 0;
    for (size_t c = 0; c < n_chunks; c++) n_reads += chunks[c].n_reads;
    s->n_reads = (uint32_t)n_reads;
    s->h_chunks.resize(n_chunks);
    s->h_reads.resize(n_reads);
    s->h_state0.resize(n_chunks);
    std::vector<uint8_t> h_tmpl;
    std::vector<uint32_t> h_opslen(n_reads);
    std::vector<uint64_t> h_homop_off(n_chunks), h_aux_off(n_chunks), h_lg_off(n_chunks);
    uint64_t tmpl_off = 0, ops_cap_off = 0, ey_off = 0, delta_off = 0, table_off = 0, raw_off = 0, row_off = 0,
             total_off = 0, edit_off = 0, feat_off = 0, cand_off = 0, aux_off = 0, lg_off = 0;
    const uint32_t H = params->gains.max_homopolymer_len;
    uint32_t rcount = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        const jtk_lc_chunk_t &ch = chunks[c];
        const uint32_t tl = (uint32_t)ch.tmpl_len;
        if (ch.read_first != rcount) return fail(JTK_ERR_INVALID_ARG, "chunks must list their reads contiguously in order");
        const uint32_t cap = tl + tl / 8 + 64;
        ChunkMeta &cm = s->h_chunks[c];
        memset(&cm, 0, sizeof cm);
        cm.chunk_id = ch.chunk_id;
        // one clustering() call: copy_num itself, or BRANCH_NUM = 4 in the split branch (mod.rs:136-142)
        cm.copy_num = ch.copy_num > JTK_MAX_COPY ? 4 : ch.copy_num;
        s->h_copy0[c] = ch.copy_num;
        if (ch.copy_num > JTK_MAX_COPY) s->has_split = true;
        // a posterior row holds up to cluster_num <= copy_num entries (merged sub-clusterings included, mod.rs:161-187)
        // (a sub-problem of the split branch is one clustering() call: its rows hold cm.copy_num entries)
        if ((extra ? cm.copy_num : ch.copy_num) > post_stride && ch.n_reads > 0)
            return fail(JTK_ERR_INVALID_ARG, "post_stride smaller than a chunk's copy_num");
        cm.n_reads = ch.n_reads;
        cm.read_first = rcount;
        cm.tmpl_cap = cap;
        cm.tmpl_off = tmpl_off;
        cm.total_off = total_off;
        const uint32_t band_width = (uint32_t)std::ceil((double)tl * params->band_frac);
        cm.radius = extra ? extra[c].radius : band_width / 2;
        cm.edit_cap = cap / 2 + 2;
        cm.edit_off = edit_off;
        cm.feat_off = feat_off;
        cm.cand_off = cand_off;
        // per_cluster_cov (mod.rs:108-111)
        const double pcc = (double)ch.n_reads / (double)ch.copy_num;
        cm.local_coverage = ch.copy_num <= 2 ? pcc : (pcc > params->haploid_coverage ? pcc : params->haploid_coverage);
        if (extra) cm.local_coverage = extra[c].local_coverage;
        if (extra) cm.take_num = extra[c].take_num;
        ChunkState &st = s->h_state0[c];
        memset(&st, 0, sizeof st);
        st.tmpl_len = tl;
        st.active = 1;
        st.k = 1;
        if (cm.radius > JTK_WIDE_MAX_RADIUS) {
            st.status = JTK_ERR_UNSUPPORTED;
        } else if (cm.radius > JTK_MAX_RADIUS) {
            s->n_wide_reads += ch.n_reads;
            if (cm.radius > s->max_wide_radius) s->max_wide_radius = cm.radius;
        }
        h_homop_off[c] = tmpl_off;
        h_aux_off[c] = aux_off;
        h_lg_off[c] = lg_off;
        for (uint32_t r = 0; r < ch.n_reads; r++) {
            const uint64_t g = rcount + r;
            const uint64_t rl = read_off[g + 1] - read_off[g], ol = ops_off[g + 1] - ops_off[g];
            ReadMeta &rm = s->h_reads[g];
            memset(&rm, 0, sizeof rm);
            rm.chunk = (uint32_t)c;
            rm.read_len = (uint32_t)rl;
            rm.ey_off = ey_off;
            rm.ops_off = ops_cap_off;
            rm.ops_cap = ((uint32_t)ol + tl / 8 + 64 + 7u) & ~7u;  // slots stay 8-byte aligned: the walkers fetch 8 ops at a time
            rm.strand = strand[g] ? 1 : 0;
            rm.delta_off = delta_off;
            rm.table_off = raw_off;  // finalize_kernel turns the row sums into the table in place
            rm.raw_off = raw_off;
            rm.row_off = row_off;
            h_opslen[g] = (uint32_t)ol;
            if (rl > s->max_read) s->max_read = (uint32_t)rl;
            ey_off += rl + 1;
            ops_cap_off += rm.ops_cap;
            delta_off += ((uint64_t)(cap + rl) >> 6) + 3;
            table_off += (uint64_t)JTK_NUM_ROW * (cap + 1);
            raw_off += (uint64_t)JTK_ACC_N * (cap + 1);
            row_off += cap + 1;
        }
        if (cap > s->max_tmpl) s->max_tmpl = cap;
        if (ch.n_reads > s->max_n) s->max_n = ch.n_reads;
        if (cm.copy_num > s->max_copy) s->max_copy = cm.copy_num;
        rcount += ch.n_reads;
        tmpl_off += cap;
        total_off += (uint64_t)JTK_NUM_ROW * (cap + 1);
        cand_off += (uint64_t)JTK_NUM_ROW * (cap + 1);
        edit_off += cm.edit_cap;
        feat_off += (uint64_t)ch.n_reads * JTK_MAX_DIM;
        aux_off += (uint64_t)3 * H * (ch.n_reads + 1) + (ch.n_reads + 1) + (JTK_MAX_COPY + 2);
        lg_off += (uint64_t)ch.n_reads * (JTK_MAX_COPY + 1);
    }
    // ---- per-base encoding.  Templates (a few MB) are recoded here; the reads and their ops -- the bulk of a stage call's input --
    //      cross the bus as the caller holds them and are validated and recoded by encode_reads_kernel (io_kernels.hip), below.
    h_tmpl.assign(tmpl_off, 0);
    {
        int bad = 0;
        for (size_t c = 0; c < n_chunks; c++) {
            const jtk_lc_chunk_t &ch = chunks[c];
            const ChunkMeta &cm = s->h_chunks[c];
            for (uint64_t p = 0; p < ch.tmpl_len; p++) {
                const int code = base_code(tmpl_bases[ch.tmpl_off + p]);
                if (code < 0) bad = 1;
                h_tmpl[cm.tmpl_off + p] = (uint8_t)(code & 3);
            }
        }
        if (bad == 1) return fail(JTK_ERR_INVALID_ARG, "non-ACGT base in a template");
    }
    s->tmpl_bytes = h_tmpl.size();
    s->ops_bytes = ops_cap_off;
    if (s->has_split) {  // the sub-problems of the split branch are encoded from these again
        s->h_read_bases.assign(read_bases, read_bases + read_off[n_reads]);
        s->h_read_off.assign(read_off, read_off + n_reads + 1);
        s->h_strand.assign(strand, strand + n_reads);
    }

    // ---- device allocation + upload
    tstart(s, -1);
    std::vector<jtk_lc_params_t> pv(1, *params);
    std::vector<HmmDev> hv = {to_dev(params->forward), to_dev(params->reverse)};
    if ((rc = dev_upload(s, s->d_params, pv))) return rc;
    if ((rc = dev_upload(s, s->d_hmm2, hv))) return rc;
    if ((rc = dev_upload(s, s->d_chunks, s->h_chunks))) return rc;
    {  // The chain kernel's launches.  Its LDS work area is sized per launch (reads x columns of features + tables), and
       // two workgroups share a CU only below 80 KiB each: chunks whose own need stays below that form class 0 (all of a
       // diploid / low-copy batch), the others class 1 (several hundred reads, or many columns).  Within a class the
       // longest chains (reads x candidate cluster counts) are dispatched first.  A class-1 combination that does not fit
       // a CU's 160 KiB loses its largest members (reported as JTK_ERR_UNSUPPORTED, like a pile-up beyond JTK_MAX_PILEUP).
        auto dims_of = [&](const ChunkMeta &cm, uint32_t *n, uint32_t *d, uint32_t *k) {
            *n = cm.n_reads;
            *k = std::min<uint32_t>(std::max<uint32_t>(cm.copy_num, 2u), JTK_MAX_COPY);
            // a chunk picks at most ROUND * max(copy_num, 2) columns (pseudo_mcmc.rs:421,527,532)
            *d = std::min<uint32_t>(JTK_MAX_DIM, 3u * std::max<uint32_t>(cm.copy_num, 2u));
        };
        std::vector<uint32_t> cls[3];
        for (uint32_t c = 0; c < s->h_chunks.size() && !polish_only; c++) {
            uint32_t n, d, k;
            dims_of(s->h_chunks[c], &n, &d, &k);
            if (s->h_chunks[c].copy_num > JTK_MAX_COPY) {
                if (s->h_state0[c].status == 0) s->h_state0[c].status = JTK_ERR_UNSUPPORTED;
                continue;
            }
            if (n > JTK_MAX_PILEUP) {  // the reference takes any depth (mod.rs:86-123): so does class 2, slowly
                cls[2].push_back(c);
                continue;
            }
            cls[mcmc_lds_bytes(n, d, k) <= 80 * 1024 ? 0 : 1].push_back(c);
        }
        std::vector<uint32_t> order;
        for (int q = 0; q < 2; q++) {
            auto &v = cls[q];
            auto need = [&](uint32_t c) {
                uint32_t n, d, k;
                dims_of(s->h_chunks[c], &n, &d, &k);
                return mcmc_lds_bytes(n, d, k);
            };
            ChainClass &cc = s->chain_class[q];
            for (;;) {
                cc = ChainClass{};
                for (uint32_t c : v) {
                    uint32_t n, d, k;
                    dims_of(s->h_chunks[c], &n, &d, &k);
                    cc.lds_n = std::max(cc.lds_n, n);
                    cc.lds_d = std::max(cc.lds_d, d);
                    cc.lds_k = std::max(cc.lds_k, k);
                }
                // class 0 is launched with its members' maxima of (reads, columns, clusters): THAT combination has to stay
                // below 80 KiB for two workgroups to share a CU, not just every member's own need
                cc.lds_bytes = v.empty() ? 0 : (uint32_t)mcmc_lds_bytes(cc.lds_n, cc.lds_d, cc.lds_k);
                if (v.empty() || cc.lds_bytes <= (q == 0 ? 80u : 160u) * 1024u) break;
                auto worst = std::max_element(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return need(a) < need(b); });
                cls[q + 1].push_back(*worst);  // a class's maxima can combine beyond one member's need: hand it over (class 1's
                                               // overflow -- a work area beyond 160 KiB -- goes to the global-memory class)
                v.erase(worst);
            }
            std::stable_sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) {
                const ChunkMeta &x = s->h_chunks[a], &y = s->h_chunks[b];
                return (uint64_t)x.n_reads * std::min<uint32_t>(x.copy_num, 4) > (uint64_t)y.n_reads * std::min<uint32_t>(y.copy_num, 4);
            });
            cc.first = (uint32_t)order.size();
            cc.count = (uint32_t)v.size();
            order.insert(order.end(), v.begin(), v.end());
        }
        {   // class 2: one workgroup per chunk, its own slice of a global workspace
            ChainClass &cc = s->chain_class[2];
            cc = ChainClass{};
            std::vector<uint64_t> ws_off;
            uint64_t ws = 0;
            for (uint32_t c : cls[2]) {
                uint32_t n, d, k;
                dims_of(s->h_chunks[c], &n, &d, &k);
                cc.lds_n = std::max(cc.lds_n, n);
                cc.lds_d = std::max(cc.lds_d, d);
                cc.lds_k = std::max(cc.lds_k, k);
                ws_off.push_back(ws);
                ws += mcmc_ws_bytes(n, d, k);
            }
            cc.first = (uint32_t)order.size();
            cc.count = (uint32_t)cls[2].size();
            order.insert(order.end(), cls[2].begin(), cls[2].end());
            if (cc.count) {
                if ((rc = dev_alloc<uint8_t>(s->d_chain_ws, ws))) return rc;
                if ((rc = dev_upload(s, s->d_chain_ws_off, ws_off))) return rc;
            }
        }
        if (order.empty()) order.push_back(0);
        if ((rc = dev_upload(s, s->d_order, order))) return rc;
    }
    if ((rc = dev_upload(s, s->d_reads, s->h_reads))) return rc;
    if ((rc = dev_alloc<ChunkState>(s->d_state, n_chunks))) return rc;
    if ((rc = dev_upload(s, s->d_tmpl_init, h_tmpl))) return rc;
    if ((rc = dev_alloc<uint8_t>(s->d_ops_init, ops_cap_off))) return rc;
    if ((rc = dev_upload(s, s->d_opslen_init, h_opslen))) return rc;
    // (+8: the polish kernels fetch templates / reads in aligned 8-byte blocks)
    if ((rc = dev_alloc<uint8_t>(s->d_tmpl0, h_tmpl.size() + 8))) return rc;
    if ((rc = dev_alloc<uint8_t>(s->d_tmpl1, h_tmpl.size() + 8))) return rc;
    if ((rc = dev_alloc<uint8_t>(s->d_ops0, ops_cap_off))) return rc;
    if ((rc = dev_alloc<uint8_t>(s->d_ops1, ops_cap_off))) return rc;
    if ((rc = dev_alloc<uint32_t>(s->d_opslen0, n_reads))) return rc;
    if ((rc = dev_alloc<uint32_t>(s->d_opslen1, n_reads))) return rc;
    if ((rc = dev_alloc<uint8_t>(s->d_ey, ey_off + 8))) return rc;
    HIP_TRY(hipMemsetAsync(s->d_ey.as<uint8_t>() + ey_off, 0, 8, s->stream));   // (the polish kernels fetch aligned 8-byte blocks)
    if ((rc = dev_alloc<uint64_t>(s->d_delta, delta_off))) return rc;
    if ((rc = dev_alloc<double>(s->d_raw, raw_off))) return rc;
    if (n_reads) {
        // The reads and their ops, as the caller holds them (ASCII bases, one op per byte, back to back), straight from the
        // caller's memory into the workspace the row sums will occupy later (296 KB per read against ~4 KB of input), then
        // validated and recoded on the device: no recoded copy on the host, no first-touch of fresh host vectors.
        const uint64_t rb0 = read_off[0], rb_bytes = read_off[n_reads] - rb0, ob0 = ops_off[0], ob_bytes = ops_off[n_reads] - ob0;
        const uint64_t o_bases = 0, o_ops = (rb_bytes + 15) & ~15ull, o_boff = o_ops + ((ob_bytes + 15) & ~15ull),
                       o_ooff = o_boff + (n_reads + 1) * 8, o_flag = o_ooff + (n_reads + 1) * 8, need = o_flag + 16;
        DevPtr tmp;   // (only when the row sums' block is too small for it: pile-ups of very short templates)
        uint8_t *stage = s->d_raw.as<uint8_t>();
        if (need > raw_off * 8) {
            if ((rc = dev_alloc<uint8_t>(tmp, need))) return rc;
            stage = tmp.as<uint8_t>();
        }
        HIP_TRY(hipMemsetAsync(stage + o_flag, 0, 16, s->stream));
        if (rb_bytes) HIP_TRY(hipMemcpyAsync(stage + o_bases, read_bases + rb0, rb_bytes, hipMemcpyHostToDevice, s->stream));
        if (ob_bytes) HIP_TRY(hipMemcpyAsync(stage + o_ops, ops + ob0, ob_bytes, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(stage + o_boff, read_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(stage + o_ooff, ops_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, s->stream));
        launch_encode_reads(s->stream, (uint32_t)n_reads, s->d_reads.as<ReadMeta>(), stage + o_bases,
                            reinterpret_cast<const uint64_t *>(stage + o_boff), stage + o_ops,
                            reinterpret_cast<const uint64_t *>(stage + o_ooff), s->d_ey.as<uint8_t>(), s->d_ops_init.as<uint8_t>(),
                            reinterpret_cast<uint32_t *>(stage + o_flag));
        uint32_t flags = 0;
        HIP_TRY(hipMemcpyAsync(&flags, stage + o_flag, 4, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));   // (also: `tmp` may go back to the pool)
        HIP_TRY(hipGetLastError());
        if (flags & 2u) return fail(JTK_ERR_INVALID_ARG, "non-ACGT base in a read");
        if (flags & 4u) return fail(JTK_ERR_INVALID_ARG, "bad op code");
    }
    if ((rc = dev_alloc<int>(s->d_rawG, row_off))) return rc;
    if ((rc = dev_alloc<double>(s->d_lk, n_reads))) return rc;
    (void)table_off;  // no buffer of its own: the tables live where the row sums were (d_raw)
    if ((rc = dev_alloc<double>(s->d_total, total_off))) return rc;
    if ((rc = dev_alloc<Edit>(s->d_edits, edit_off))) return rc;
    if ((rc = dev_alloc<uint32_t>(s->d_newlen, n_chunks))) return rc;
    if ((rc = dev_alloc<uint32_t>(s->d_counter, 4))) return rc;
    HIP_TRY(hipMemsetAsync(s->d_counter.p, 0, 4 * sizeof(uint32_t), s->stream));  // once: the ticket counters are never reset
    if ((rc = dev_alloc<uint32_t>(s->d_nactive, JTK_NACTIVE_SLOTS))) return rc;
    if ((rc = dev_alloc<uint32_t>(s->d_chain_split, 2 * n_chunks + 8))) return rc;
    if (!polish_only) {  // the second stream (the chain's general kernel beside its light one) only where the hardware queues
                         // are there for it: the host asked for >= 8 queues, or the variable was absent when this library was
                         // loaded and jtk_lc_default_hw_queues set it (a host that initialised HIP BEFORE loading the library
                         // and never set the variable runs on 4 queues: it says so with JTK_LC_SIDE_STREAM=0)
        const char *force = getenv("JTK_LC_SIDE_STREAM");
        const bool on = force ? atoi(force) != 0 : (g_queues_by_library || g_host_queues >= 8);
        if (on) {
            // experiment (JTK_CHAIN_CUS=n): the chain's two kernels on streams confined to n CUs, so that their long-lived
            // waves do not take wave slots all over the device (the light kernel then runs on `chain_main`, the general one on `side`)
            static const int chain_cus = getenv("JTK_CHAIN_CUS") ? atoi(getenv("JTK_CHAIN_CUS")) : 0;
            if (chain_cus > 0) {
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, device));
                const int n_cu = prop.multiProcessorCount;
                std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
                static const int spread = getenv("JTK_CHAIN_CUS_SPREAD") ? atoi(getenv("JTK_CHAIN_CUS_SPREAD")) : 1;
                for (int k = 0; k < chain_cus && k < n_cu; k++) {
                    const int cu = spread ? (int)((int64_t)k * n_cu / chain_cus) : k;   // spread over the device, or the first n
                    mask[cu / 32] |= 1u << (cu % 32);
                }
                HIP_TRY(hipExtStreamCreateWithCUMask(&s->side, (uint32_t)mask.size(), mask.data()));
                HIP_TRY(hipExtStreamCreateWithCUMask(&s->chain_main, (uint32_t)mask.size(), mask.data()));
                HIP_TRY(hipEventCreateWithFlags(&s->ev_chain[2], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&s->ev_chain[3], hipEventDisableTiming));
            } else
            HIP_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&s->ev_chain[0], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&s->ev_chain[1], hipEventDisableTiming));
        }
    }
    if (!polish_only) {  // the filter's and the chain's workspaces
    if ((rc = dev_alloc<uint16_t>(s->d_homop, tmpl_off))) return rc;
    if ((rc = dev_upload(s, s->d_homop_off, h_homop_off))) return rc;
    if ((rc = dev_alloc<double>(s->d_aux, aux_off))) return rc;
    if ((rc = dev_upload(s, s->d_aux_off, h_aux_off))) return rc;
    if ((rc = dev_alloc<double>(s->d_cand, cand_off))) return rc;
    if ((rc = dev_alloc<uint32_t>(s->d_list, cand_off))) return rc;
    if ((rc = dev_alloc<uint8_t>(s->d_sel, cand_off))) return rc;
    if ((rc = dev_alloc<double>(s->d_feat, feat_off))) return rc;
    if ((rc = dev_alloc<uint32_t>(s->d_vtype, (uint64_t)2 * n_chunks * JTK_MAX_DIM))) return rc;
    if ((rc = dev_alloc<uint32_t>(s->d_pos, (uint64_t)n_chunks * JTK_MAX_DIM))) return rc;
    if ((rc = dev_alloc<uint32_t>(s->d_label, n_reads))) return rc;
    if ((rc = dev_alloc<double>(s->d_post, (uint64_t)n_reads * post_stride))) return rc;
    if ((rc = dev_alloc<double>(s->d_lg, lg_off))) return rc;
    if ((rc = dev_upload(s, s->d_lg_off, h_lg_off))) return rc;
    }
    if (extra) {
        std::vector<uint64_t> h_rng(4 * n_chunks);
        for (size_t c = 0; c < n_chunks; c++) memcpy(&h_rng[4 * c], extra[c].rng, 32);
        if ((rc = dev_upload(s, s->d_rng, h_rng))) return rc;
        s->resume_rng = true;
    }
    // forward scratch: one stripe per resident wave
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        // resident waves of phmm_kernel: 3 per SIMD by registers, 11 per CU by its ~14 KB of LDS
        // resident waves per CU: 3 per SIMD by registers (JTK_PHMM_WAVES), and what 160 KiB of LDS hold
        uint32_t per_cu = (uint32_t)((160u * 1024u) / phmm_lds_bytes(s->max_tmpl, s->max_read));
        if (per_cu > 12) per_cu = 12;
        if (per_cu < 1) per_cu = 1;
        if (const char *e = getenv("JTK_PHMM_WAVES_PER_CU")) per_cu = (uint32_t)atoi(e);  // tuning experiments only
        const uint32_t want = (uint32_t)prop.multiProcessorCount * per_cu;
        s->n_waves = n_reads < want ? (uint32_t)n_reads : want;
        if (s->n_waves == 0) s->n_waves = 1;
        // one stripe per wave the DEVICE can hold (3 per SIMD by registers = 12 per CU; 16 leaves room): the set is shared
        const uint64_t stride = (uint64_t)(s->max_tmpl + s->max_read + 8 + JTK_SCRATCH_GUARD) * 64 * 2;  // doubles
        const uint32_t n_stripes = (uint32_t)prop.multiProcessorCount * 16;
        static const bool private_set = getenv("JTK_STRIPE_SHARED") && atoi(getenv("JTK_STRIPE_SHARED")) == 0;  // experiments
        std::lock_guard<std::mutex> lock(g_stripe_mutex);
        std::shared_ptr<StripePool> mine;  // JTK_STRIPE_SHARED=0: a set of this session's own, as before round 3
        // A session that launches far fewer waves than the device holds (one pile-up's modification table, the five training
        // pile-ups of the model refit) neither needs nor may grow the device's set: n_waves stripes of its own (a 60-read call
        // holds 0.3 GB, not the 19 GB of 4,096 stripes).  It uses the device's set if one of sufficient stride is already there.
        const bool small = (uint64_t)s->n_waves * 4u <= n_stripes &&
                           !(device < JTK_POOL_DEVICES && g_stripes[device] && g_stripes[device]->stride >= stride);
        std::shared_ptr<StripePool> &cur = (private_set || small || device >= JTK_POOL_DEVICES) ? mine : g_stripes[device];
        const uint32_t n_want = (private_set || small) ? s->n_waves : n_stripes;
        if (!cur || cur->stride < stride || cur->n < n_want) {
            auto p = std::make_shared<StripePool>();
            p->stride = std::max<uint64_t>(stride, cur ? cur->stride : 0);
            p->n = std::max<uint32_t>(n_want, cur ? cur->n : 0);
            cur.reset();  // its blocks go back to the block cache first (if no session holds it any more)
            if ((rc = dev_alloc<double>(p->mem, p->stride * p->n))) return rc;
            if ((rc = dev_alloc<uint32_t>(p->owner, p->n))) return rc;
            // every stripe free.  On the session's own stream, waited for under the lock -- NOT hipMemset: that is an operation of
            // the null stream, which then holds a hardware queue of its own for the life of the process; with four slices x two
            // streams on eight queues one slice's second stream then shared a queue with its first, and that slice's two chain
            // kernels ran one after the other (serial chain time 1,340 -> 1,490 ms per pass until this was found)
            HIP_TRY(hipMemsetAsync(p->owner.p, 0, (size_t)p->n * sizeof(uint32_t), s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            cur = p;
        }
        s->stripes = cur;
    }
    {  // narrow bands: two reads of a chunk per wave (phmm_pair.hip); JTK_PHMM_PAIR=0 keeps them on phmm_kernel
        static const bool pair_on = []() {
            const char *e = getenv("JTK_PHMM_PAIR");
            return !(e && e[0] == '0');
        }();
        std::vector<uint32_t> items;
        if (pair_on)
            for (size_t c = 0; c < n_chunks; c++) {
                const ChunkMeta &cm = s->h_chunks[c];
                if (cm.radius > JTK_PAIR_MAX_RADIUS || s->h_state0[c].status != 0) continue;
                const uint32_t voters = cm.take_num ? std::min(cm.take_num, cm.n_reads) : cm.n_reads;
                for (uint32_t r = 0; r + 1 < voters; r += 2) items.push_back(cm.read_first + r);
                if (voters & 1u) items.push_back((cm.read_first + voters - 1) | 0x80000000u);
            }
        s->n_pair_items = (uint32_t)items.size();
        if (s->n_pair_items) {
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, device));
            const size_t pl = phmm_pair_lds_bytes(s->max_tmpl, s->max_read);
            if (pl > 160 * 1024) return fail(JTK_ERR_UNSUPPORTED, "template + reads too long for the LDS staging of phmm_pair_kernel");
            uint32_t per_cu = (uint32_t)((160u * 1024u) / pl);
            if (per_cu > 8) per_cu = 8;  // two waves per SIMD by registers
            s->n_pair_waves = std::min<uint32_t>(std::min<uint32_t>(s->n_pair_items, (uint32_t)prop.multiProcessorCount * per_cu),
                                                 s->n_waves);  // shares phmm_kernel's scratch stripes (launched after it)
            if ((rc = dev_upload(s, s->d_pair_items, items))) return rc;
        }
    }
    if (s->n_wide_reads) {
        const size_t wl = phmm_wide_lds_bytes(s->max_tmpl, s->max_read, s->max_wide_radius);
        if (wl > 160 * 1024) return fail(JTK_ERR_UNSUPPORTED, "template + read too long for the LDS staging of phmm_wide_kernel");
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        uint32_t per_cu = (uint32_t)((160u * 1024u) / wl);
        if (per_cu > 4) per_cu = 4;
        s->wide_stride = phmm_wide_scratch_doubles(s->max_tmpl, s->max_read, s->max_wide_radius);
        uint64_t want = (uint64_t)prop.multiProcessorCount * per_cu;
        const uint64_t budget = (48ull << 30) / (s->wide_stride * 8);  // at most 48 GB of forward tables in flight
        if (want > budget) want = budget ? budget : 1;
        s->n_wide_waves = (uint32_t)std::min<uint64_t>(s->n_wide_reads, want);
        if ((rc = dev_alloc<double>(s->d_wide_scratch, s->wide_stride * s->n_wide_waves))) return rc;
        if ((rc = dev_alloc<uint32_t>(s->d_wide_counter, 4))) return rc;
        HIP_TRY(hipMemsetAsync(s->d_wide_counter.p, 0, 4 * sizeof(uint32_t), s->stream));
    }
    // set 0 = the batch as uploaded (never written), sets 1 / 2 = what the polish rounds write (device_common.h DevBufs)
    s->bufs.tmpl[0] = s->d_tmpl_init.as<uint8_t>();
    s->bufs.tmpl[1] = s->d_tmpl0.as<uint8_t>();
    s->bufs.tmpl[2] = s->d_tmpl1.as<uint8_t>();
    s->bufs.ops[0] = s->d_ops_init.as<uint8_t>();
    s->bufs.ops[1] = s->d_ops0.as<uint8_t>();
    s->bufs.ops[2] = s->d_ops1.as<uint8_t>();
    s->bufs.ops_len[0] = s->d_opslen_init.as<uint32_t>();
    s->bufs.ops_len[1] = s->d_opslen0.as<uint32_t>();
    s->bufs.ops_len[2] = s->d_opslen1.as<uint32_t>();
    if ((rc = dev_upload(s, s->d_state0, s->h_state0))) return rc;
    tstop(s);
    HIP_TRY(hipStreamSynchronize(s->stream));
    {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, s->timers.back().a, s->timers.back().b);
        memset(&g_timing, 0, sizeof g_timing);
        g_timing.h2d_ms = ms;
    }
    const size_t lds = phmm_lds_bytes(s->max_tmpl, s->max_read);
    if (lds > 160 * 1024) return fail(JTK_ERR_UNSUPPORTED, "template + read too long for the LDS staging of phmm_kernel");
    *out = guard.release();
    return 0;
}

int jtk_lc_session_create(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                          const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                          const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand,
                          uint32_t post_stride, int device, jtk_lc_session_t **out) {
    return session_create_ex(params, n_chunks, chunks, tmpl_bases, read_bases, read_off, ops, ops_off, strand,
                             post_stride, device, nullptr, 3 /* mod.rs:105 */, out);
}

static int run_split(jtk_lc_session_t *s);

// one pass of the kernel sequence over the resident batch
static int run_batch(jtk_lc_session_t *s, int skip_polish) {
    HIP_TRY(hipSetDevice(s->device));
    PhaseHold phase(s->device);   // the pair-HMM gate: held until the chain kernels are queued (or the pass ends)
    const double h2d = g_timing.h2d_ms;
    memset(&g_timing, 0, sizeof g_timing);
    g_timing.h2d_ms = h2d;
    for (auto &t : s->timers) {
        if (t.a) (void)hipEventDestroy(t.a);
        if (t.b) (void)hipEventDestroy(t.b);
    }
    s->timers.clear();
    hipStream_t st = s->stream;
    const ReadMeta *reads = s->d_reads.as<ReadMeta>();
    const ChunkMeta *chunks = s->d_chunks.as<ChunkMeta>();
    ChunkState *state = s->d_state.as<ChunkState>();
    const HmmDev *hmm2 = s->d_hmm2.as<HmmDev>();
    hipEvent_t ev0, ev1;
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, st));
    // reset the mutable state: every chunk back to the batch as uploaded (buffer set 0 is never written, so there is
    // nothing to copy) and the per-round counters to zero -- one small kernel, no blits
    launch_reset_pass(st, s->n_chunks, state, s->d_state0.as<ChunkState>(), s->d_nactive.as<uint32_t>(), JTK_NACTIVE_SLOTS);

    tstart(s, JTK_K_POLISH);
    launch_band_prep(st, s->n_reads, reads, chunks, state, s->bufs, s->d_delta.as<uint64_t>(), 0, s->max_tmpl, s->max_read);
    tstop(s);
    // The host has to learn when every chunk has converged, but the device never waits for it: round r+1 is queued BEFORE the
    // host looks at round r's counter (a round without active chunks does nothing: every kernel of a round >= 1 skips the
    // chunks that are not active), so a pass costs at most one empty round instead of a host round trip per round.
    if (!s->h_nactive) {
        s->h_nactive = pinned_page_take();
        if (!s->h_nactive) return fail(JTK_ERR_ALLOC, "hipHostMalloc failed");
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->h_nactive_dev), s->h_nactive, 0));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_round[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_round[1], hipEventDisableTiming));
    }
    // Round 6: the variant filter takes its column statistics and the picked columns' entries straight from the row sums
    // (column_filter_fused_kernel, filter_kernels.hip), so a clustering pass never materialises the N x 14(L+1) table --
    // finalize_kernel runs only where the table itself is the product (window polishing keeps its final-pass call; the
    // modification-table entry point has its own) or with JTK_FILTER_FUSED=0 (rounds 1-5, kept for differential runs).
    static const bool filter_fused = !(getenv("JTK_FILTER_FUSED") && atoi(getenv("JTK_FILTER_FUSED")) == 0);
    const bool need_tables = s->polish_only || !filter_fused || s->max_n > 65535u;  // (the fused filter counts in 16-bit fields)
    const int max_rounds = skip_polish ? 1 : JTK_POLISH_MAX_ROUNDS + 1;
    for (int round = 0; round < max_rounds; round++) {
        const int only_active = round > 0;
        const int final_pass = skip_polish || round == JTK_POLISH_MAX_ROUNDS;
        tstart(s, JTK_K_PHMM);
        launch_phmm(st, s->n_reads, reads, chunks, state, s->bufs, s->d_ey.as<uint8_t>(), s->d_delta.as<uint64_t>(),
                    hmm2, s->stripes->set(), s->n_waves, s->d_counter.as<uint32_t>(), &s->tk_phmm,
                    s->d_raw.as<double>(), s->d_rawG.as<int>(), s->d_lk.as<double>(), s->max_tmpl, s->max_read,
                    only_active, s->n_pair_items ? JTK_PAIR_MAX_RADIUS : 0);
        if (s->n_pair_items)
            launch_phmm_pair(st, s->n_pair_items, s->d_pair_items.as<uint32_t>(), reads, chunks, state, s->bufs,
                             s->d_ey.as<uint8_t>(), s->d_delta.as<uint64_t>(), hmm2, s->stripes->set(),
                             s->n_pair_waves, s->d_counter.as<uint32_t>() + 1, &s->tk_pair, s->d_raw.as<double>(), s->d_rawG.as<int>(),
                             s->d_lk.as<double>(), s->max_tmpl, s->max_read, only_active);
        if (s->n_wide_reads)
            launch_phmm_wide(st, s->n_reads, reads, chunks, state, s->bufs, s->d_ey.as<uint8_t>(), s->d_delta.as<uint64_t>(),
                             hmm2, s->d_wide_scratch.as<double>(), s->wide_stride, s->n_wide_waves,
                             s->d_wide_counter.as<uint32_t>(), &s->tk_wide, s->d_raw.as<double>(), s->d_rawG.as<int>(),
                             s->d_lk.as<double>(), s->max_tmpl, s->max_read, only_active, s->max_wide_radius);
        // A polish round needs the column totals of the active chunks, not their per-read tables: the totals come straight from
        // the row sums (sum_final_kernel), and only a chunk that turns out to have converged -- no edit selected -- gets its
        // table, once, from the row sums it still holds.  The last pass materialises the tables of whatever is still active.
        if (final_pass) {
            if (need_tables)
                launch_finalize(st, s->n_reads, reads, chunks, state, hmm2, s->d_raw.as<double>(), s->d_rawG.as<int>(),
                                s->d_lk.as<double>(), s->max_tmpl, only_active);
        } else
            launch_sum_final(st, s->n_chunks, reads, chunks, state, hmm2, s->d_raw.as<double>(), s->d_rawG.as<int>(),
                             s->d_lk.as<double>(), s->d_total.as<double>(), s->max_tmpl);
        tstop(s);
        tstart(s, JTK_K_POLISH);
        launch_polish_round(st, s->n_chunks, s->n_reads, reads, chunks, state, s->bufs, s->d_ey.as<uint8_t>(),
                            s->d_raw.as<double>(), s->d_total.as<double>(), s->d_edits.as<Edit>(),
                            s->d_newlen.as<uint32_t>(), s->max_tmpl, s->ignore_edge, final_pass,
                            s->d_nactive.as<uint32_t>() + round, s->h_nactive_dev + round);
        if (!final_pass)
            launch_band_prep(st, s->n_reads, reads, chunks, state, s->bufs, s->d_delta.as<uint64_t>(), 1, s->max_tmpl, s->max_read);
        tstop(s);
        if (!final_pass && !s->polish_only && need_tables) {  // the chunks that converged in this round: their tables, for the variant search
            tstart(s, JTK_K_PHMM);
            launch_finalize(st, s->n_reads, reads, chunks, state, hmm2, s->d_raw.as<double>(), s->d_rawG.as<int>(),
                            s->d_lk.as<double>(), s->max_tmpl, 0, round);
            tstop(s);
        }
        if (final_pass) break;
        // commit_kernel has stored the round's count in h_nactive[round] itself (mapped pinned memory): no read-back copy
        HIP_TRY(hipEventRecord(s->ev_round[round & 1], st));
        if (round >= 1) {
            HIP_TRY(hipEventSynchronize(s->ev_round[(round - 1) & 1]));
            if (s->h_nactive[round - 1] == 0) break;  // round `round` was already queued and finds nothing to do
        }
    }
    if (!s->polish_only) {
    tstart(s, JTK_K_FILTER);
    launch_filter(st, s->n_chunks, reads, chunks, state, s->bufs, s->d_params.as<jtk_lc_params_t>(),
                  s->d_raw.as<double>(), s->d_homop.as<uint16_t>(), s->d_homop_off.as<uint64_t>(),
                  s->d_aux.as<double>(), s->d_aux_off.as<uint64_t>(), s->d_cand.as<double>(), s->d_list.as<uint32_t>(),
                  s->d_sel.as<uint8_t>(), s->d_feat.as<double>(), s->d_vtype.as<uint32_t>(), s->d_pos.as<uint32_t>(),
                  s->max_tmpl, hmm2, s->d_rawG.as<int>(), s->d_lk.as<double>(), need_tables ? 0 : 1);
    tstop(s);
    phase.release();   // everything that fills the device is queued: the next batch may start its rounds
    tstart(s, JTK_K_MCMC);
    int mcmc_rc = 0;
    hipStream_t st_main = st;
    if (s->chain_main) {   // (experiment) the chain on its CU-masked stream: fork here, join after the launches
        HIP_TRY(hipEventRecord(s->ev_chain[2], st));
        HIP_TRY(hipStreamWaitEvent(s->chain_main, s->ev_chain[2], 0));
        st = s->chain_main;
    }
    // measurement only (scripts/overlap_probe_r6.sh): a pass without its chain kernels -- what the chain's share of the CUs' LDS
    // and issue slots costs the pair-HMM family when slices overlap.  Labels / posteriors are then whatever the buffers held.
    static const bool x_nochain = getenv("JTK_X_NOCHAIN") != nullptr;
    for (int j = 0; j < 2; j++) {
        const ChainClass &cc = s->chain_class[j];
        if (cc.count == 0 || mcmc_rc != 0 || x_nochain) continue;
        // the class's own stretch of the split scratch: 2 + 2 * count words from 2 * first + 4 * j
        mcmc_rc = launch_mcmc(st, cc.count, chunks, state, s->d_params.as<jtk_lc_params_t>(), s->d_feat.as<double>(),
                              s->d_vtype.as<uint32_t>(), nullptr, 0, s->d_label.as<uint32_t>(), s->d_post.as<double>(),
                              s->post_stride, s->d_lg.as<double>(), s->d_lg_off.as<uint64_t>(), cc.lds_n, cc.lds_d, cc.lds_k,
                              s->resume_rng ? s->d_rng.as<uint64_t>() : nullptr, s->d_order.as<uint32_t>() + cc.first,
                              s->d_chain_split.as<uint32_t>() + 2 * cc.first + 4 * j, s->side, s->ev_chain[0], s->ev_chain[1]);
    }
    if (s->chain_class[2].count && mcmc_rc == 0 && !x_nochain) {
        const ChainClass &cc = s->chain_class[2];
        mcmc_rc = launch_mcmc_huge(st, cc.count, chunks, state, s->d_params.as<jtk_lc_params_t>(), s->d_feat.as<double>(),
                                   s->d_vtype.as<uint32_t>(), nullptr, 0, s->d_label.as<uint32_t>(), s->d_post.as<double>(),
                                   s->post_stride, s->d_lg.as<double>(), s->d_lg_off.as<uint64_t>(), cc.lds_n, cc.lds_d, cc.lds_k,
                                   s->resume_rng ? s->d_rng.as<uint64_t>() : nullptr, s->d_order.as<uint32_t>() + cc.first,
                                   s->d_chain_ws.as<uint8_t>(), s->d_chain_ws_off.as<uint64_t>());
    }
    if (s->chain_main) {
        HIP_TRY(hipEventRecord(s->ev_chain[3], s->chain_main));
        HIP_TRY(hipStreamWaitEvent(st_main, s->ev_chain[3], 0));
        st = st_main;
    }
    tstop(s);
    if (mcmc_rc != 0) {
        (void)hipStreamSynchronize(st);
        return fail(JTK_ERR_INTERNAL, "the chain kernel could not be launched (jump table upload failed)");
    }
    s->ran = true;
    s->ran_fused = !need_tables;
    }  // !polish_only
    HIP_TRY(hipEventRecord(ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    g_timing.total_ms = ms;
    for (int j = 0; j < 2; j++) g_timing.chain_lds_bytes[j] = s->chain_class[j].count ? s->chain_class[j].lds_bytes : 0;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    for (auto &t : s->timers) {
        if (t.kind < 0 || t.kind >= JTK_K_COUNT) continue;
        float k = 0;
        (void)hipEventElapsedTime(&k, t.a, t.b);
        g_timing.kernel_ms[t.kind] += k;
        g_timing.kernel_launches[t.kind] += 1;
    }
    return 0;
}

int jtk_lc_session_run(jtk_lc_session_t *s, int skip_polish) {
    g_last_error.clear();
    if (!s) return fail(JTK_ERR_INVALID_ARG, "null session");
    int rc = run_batch(s, skip_polish);
    if (rc == 0 && s->has_split) rc = run_split(s);
    return rc;
}

// ---- the reference's trace! rows of one chunk (include/jtk_lc.h: jtk_lc_session_trace)
namespace {
void trace_row(std::string &out, const char *fmt, ...) {
    char line[512];
    va_list ap;
    va_start(ap, fmt);
    int m = vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    if (m < 0) return;
    out.append(line, std::min<size_t>((size_t)m, sizeof line - 1));
    out.push_back('\n');
}
std::string fx(double x, int prec) {  // Rust's {x:.N}: the correctly rounded decimal, as printf's; NaN is "NaN"
    if (x != x) return "NaN";
    char b[400];
    snprintf(b, sizeof b, "%.*f", prec, x);
    return b;
}
}  // namespace

int jtk_lc_session_trace(jtk_lc_session_t *s, size_t chunk, char *text, size_t cap, size_t *len) {
    g_last_error.clear();
    if (!s || !len || (cap && !text)) return fail(JTK_ERR_INVALID_ARG, "null argument");
    *len = 0;
    if (chunk >= s->n_chunks) return fail(JTK_ERR_INVALID_ARG, "no such chunk in the session");
    if (s->polish_only || s->features_only || !s->ran)
        return fail(JTK_ERR_INVALID_ARG, "jtk_lc_session_trace needs a session whose last jtk_lc_session_run clustered its chunks");
    if (s->has_split)  // (run_split re-uses the session's buffers for the sub-problems of clustering_recursive, mod.rs:125-189)
        return fail(JTK_ERR_UNSUPPORTED, "jtk_lc_session_trace: the session holds a chunk of copy number >= 8 (clustering_recursive)");
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = s->stream;
    const uint32_t ci = (uint32_t)chunk;
    const ChunkMeta &cm = s->h_chunks[ci];
    ChunkState cs;
    HIP_TRY(hipMemcpyAsync(&cs, s->d_state.as<ChunkState>() + ci, sizeof cs, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (cs.status != 0) return fail(cs.status, "jtk_lc_session_trace: the chunk failed in the run");
    std::string out;
    if (cm.copy_num < 2) {  // clustering() returns before anything is logged (pseudo_mcmc.rs:86-88)
        *len = 0;
        return JTK_OK;
    }
    const uint32_t n = cm.n_reads, L = cs.tmpl_len, cols = JTK_NUM_ROW * (L + 1);
    // ---- the pick once more, with its trace arrays
    DevPtr d_tr, d_cnt, d_order1, d_ws, d_ws_off, d_trace;
    int rc;
    if ((rc = dev_alloc<uint32_t>(d_tr, 2 + JTK_TRACE_MAX_PICKS))) return rc;
    if ((rc = dev_alloc<uint32_t>(d_cnt, cols))) return rc;
    HIP_TRY(hipMemsetAsync(d_tr.p, 0, (2 + JTK_TRACE_MAX_PICKS) * sizeof(uint32_t), st));
    launch_pick_trace(st, ci, s->d_reads.as<ReadMeta>(), s->d_chunks.as<ChunkMeta>(), s->d_state.as<ChunkState>(),
                      s->d_params.as<jtk_lc_params_t>(), s->d_raw.as<double>(), s->d_homop.as<uint16_t>(), s->d_homop_off.as<uint64_t>(),
                      s->d_cand.as<double>(), s->d_list.as<uint32_t>(), s->d_sel.as<uint8_t>(), s->d_feat.as<double>(),
                      s->d_vtype.as<uint32_t>(), s->d_pos.as<uint32_t>(), s->d_hmm2.as<HmmDev>(), s->d_rawG.as<int>(), s->d_lk.as<double>(),
                      s->ran_fused ? 1 : 0, d_tr.as<uint32_t>(), d_cnt.as<uint32_t>());
    std::vector<uint32_t> tr(2 + JTK_TRACE_MAX_PICKS);
    HIP_TRY(hipMemcpyAsync(tr.data(), d_tr.p, tr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&cs, s->d_state.as<ChunkState>() + ci, sizeof cs, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    const uint32_t np = std::min(tr[0], cols), n_picks = std::min<uint32_t>(tr[1], JTK_TRACE_MAX_PICKS), D = cs.dim;
    std::vector<uint32_t> list(np), cnt(np), pos(JTK_MAX_DIM);
    std::vector<double> cand(cols), feat((size_t)n * D);
    if (np) {
        HIP_TRY(hipMemcpyAsync(list.data(), s->d_list.as<uint32_t>() + cm.cand_off, np * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(cnt.data(), d_cnt.p, np * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipMemcpyAsync(cand.data(), s->d_cand.as<double>() + cm.cand_off, cols * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(pos.data(), s->d_pos.as<uint32_t>() + (uint64_t)ci * JTK_MAX_DIM, JTK_MAX_DIM * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, st));
    if (!feat.empty())
        HIP_TRY(hipMemcpyAsync(feat.data(), s->d_feat.as<double>() + cm.feat_off, feat.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    auto diff_letter = [](uint32_t row) { return row < 4 ? "S" : (row < 8 + JTK_COPY_SIZE ? "I" : "D"); };  // pos_to_bp_and_difftype :168-178
    trace_row(out, "TOTAL\t%u", np);                                                                       // :467
    for (uint32_t i = 0; i < np; i++)                                                                       // :468-472
        trace_row(out, "CAND\t%u\t%u\t%s\t%u", list[i] / JTK_NUM_ROW, list[i] % JTK_NUM_ROW, fx(cand[list[i]], 1).c_str(), cnt[i]);
    for (uint32_t p = 0; p < n_picks; p++) {                                                                // :537-539
        const uint32_t col = tr[2 + p] < np ? list[tr[2 + p]] : 0;
        trace_row(out, "PICK\t%u\t%s\t%s", col / JTK_NUM_ROW, diff_letter(col % JTK_NUM_ROW), fx(cand[col], 3).c_str());
    }
    for (uint32_t d = 0; d < D; d++) {                                                                      // :122-127
        double sum = 0.0;
        for (uint32_t r = 0; r < n; r++) {
            const double x = feat[(size_t)r * D + d];
            sum += x != x ? 0.0 : (x > 0.0 ? x : 0.0);  // f64::max(x, 0) ignores NaN
        }
        trace_row(out, "DUMP\t%u\t%u\t%u\t%s\t%s", d, pos[d] / JTK_NUM_ROW, pos[d] % JTK_NUM_ROW, fx(cand[pos[d]], 1).c_str(),
                  fx(sum, 1).c_str());
    }
    // ---- cluster_filtered_variants once more, with its records (:213-274); its early return logs nothing (:221-225)
    if (!(D == 0 || n <= cm.copy_num)) {
        const uint32_t k = std::min<uint32_t>(std::max<uint32_t>(cm.copy_num, 2u), JTK_MAX_COPY);
        const uint32_t dmax = std::min<uint32_t>(JTK_MAX_DIM, 3u * std::max<uint32_t>(cm.copy_num, 2u));
        std::vector<uint32_t> order1(1, ci);
        std::vector<uint64_t> ws_off1(1, 0);
        if ((rc = dev_upload(s, d_order1, order1))) return rc;
        if ((rc = dev_upload(s, d_ws_off, ws_off1))) return rc;
        if ((rc = dev_alloc<uint8_t>(d_ws, mcmc_ws_bytes(n, dmax, k)))) return rc;
        const size_t tn = mcmc_trace_doubles(n);
        if ((rc = dev_alloc<double>(d_trace, tn))) return rc;
        HIP_TRY(hipMemsetAsync(d_trace.p, 0, tn * sizeof(double), st));
        if (launch_mcmc_trace(st, s->d_chunks.as<ChunkMeta>(), s->d_state.as<ChunkState>(), s->d_params.as<jtk_lc_params_t>(),
                              s->d_feat.as<double>(), s->d_vtype.as<uint32_t>(), s->d_label.as<uint32_t>(), s->d_post.as<double>(),
                              s->post_stride, s->d_lg.as<double>(), s->d_lg_off.as<uint64_t>(), n, dmax, k, d_order1.as<uint32_t>(),
                              d_ws.as<uint8_t>(), d_ws_off.as<uint64_t>(), d_trace.as<double>()) != 0) {
            (void)hipStreamSynchronize(st);
            return fail(JTK_ERR_INTERNAL, "the chain kernel could not be launched (jump table upload failed)");
        }
        std::vector<double> tv(tn);
        HIP_TRY(hipMemcpyAsync(tv.data(), d_trace.p, tn * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&cs, s->d_state.as<ChunkState>() + ci, sizeof cs, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipGetLastError());
        if (cs.status != 0) return fail(cs.status, "jtk_lc_session_trace: the chunk failed when its chain ran again");
        trace_row(out, "RANGE\t%u..=%u", (unsigned)tv[0], (unsigned)tv[1]);                                 // :236
        const uint32_t nrec = std::min<uint32_t>((uint32_t)tv[2], 8u);
        for (uint32_t r = 0; r < nrec; r++) {
            const double *rec = tv.data() + 8 + 16 * r;
            const unsigned kk = (unsigned)rec[0];
            trace_row(out, "LK\t%u\t%s", kk, fx(rec[1], 3).c_str());                                        // :250
            trace_row(out, "LK\t%u\t%s\t%s\t%u", kk, fx(rec[1], 3).c_str(), fx(rec[2], 3).c_str(), (unsigned)rec[3]);  // :256
            if (rec[4] != 0.0) {                                                                            // :258-262
                std::string c = "COUNTS\t[";
                for (unsigned q = 0; q < kk && q < JTK_MAX_COPY; q++) c += (q ? ", " : "") + std::to_string((unsigned)rec[5 + q]);
                trace_row(out, "%s]", c.c_str());
            }
        }
    }
    *len = out.size();
    if (out.size() > cap) return fail(JTK_ERR_INVALID_ARG, "jtk_lc_session_trace: the text needs " + std::to_string(out.size()) + " bytes");
    if (!out.empty()) memcpy(text, out.data(), out.size());
    return JTK_OK;
}

// include/jtk_lc_debug.h: per chunk, the cycles of its chain and the proposals that could not be stepped over
int jtk_lc_debug_chain_profile(jtk_lc_session_t *s, uint64_t *cycles, uint32_t *events) {
    g_last_error.clear();
    if (!s) return fail(JTK_ERR_INVALID_ARG, "null session");
    HIP_TRY(hipSetDevice(s->device));
    std::vector<ChunkState> state(s->n_chunks);
    HIP_TRY(hipMemcpyAsync(state.data(), s->d_state.p, state.size() * sizeof(ChunkState), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (uint32_t c = 0; c < s->n_chunks; c++) {
        if (cycles) cycles[c] = state[c].chain_cycles;
        if (events) events[c] = state[c].chain_events;
    }
    return JTK_OK;
}

// A fetch in two halves, so that the slices of a one-shot call can write their variable-length outputs straight into the
// caller's arrays: fetch_begin copies the fixed-size results (labels, posteriors, per-chunk records) and learns the lengths of the
// variable ones; the caller then says where this session's consensus / ops start (one-shot slices: after the previous slice's),
// and fetch_finish packs them on the device (gather_kernel) and copies exactly those bytes to their final place.  Until round 5
// every buffer set some chunk's result lived in was downloaded whole into fresh host vectors (up to 3 x the ops + templates of the
// batch) and unpacked base by base on the host.
namespace {
struct FetchPlan {
    std::vector<ChunkState> state;
    std::vector<uint32_t> len;      // n_reads ops lengths, then n_chunks consensus lengths
    uint64_t cons_total = 0, ops_total = 0;
    bool want_cons = false, want_ops = false;
    int any_fail = 0;
    hipEvent_t ev0 = nullptr;
};

int fetch_begin(jtk_lc_session_t *s, FetchPlan &pl, uint32_t *label, double *log_post, jtk_lc_result_t *result, bool want_cons,
                bool want_ops) {
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = s->stream;
    HIP_TRY(hipEventCreate(&pl.ev0));
    HIP_TRY(hipEventRecord(pl.ev0, st));
    pl.want_cons = want_cons;
    pl.want_ops = want_ops;
    pl.state.resize(s->n_chunks);
    HIP_TRY(hipMemcpyAsync(pl.state.data(), s->d_state.p, pl.state.size() * sizeof(ChunkState), hipMemcpyDeviceToHost, st));
    if (want_cons || want_ops) {
        if (!s->d_out_len.p) {
            int rc;
            if ((rc = dev_alloc<uint32_t>(s->d_out_len, (size_t)s->n_reads + s->n_chunks))) return rc;
            if ((rc = dev_alloc<uint64_t>(s->d_out_off, (size_t)s->n_reads + s->n_chunks + 2))) return rc;
        }
        pl.len.resize((size_t)s->n_reads + s->n_chunks);
        launch_out_len(st, s->n_reads, s->n_chunks, s->d_reads.as<ReadMeta>(), s->d_state.as<ChunkState>(), s->bufs,
                       s->d_out_len.as<uint32_t>(), s->d_out_len.as<uint32_t>() + s->n_reads);
        if (!pl.len.empty())
            HIP_TRY(hipMemcpyAsync(pl.len.data(), s->d_out_len.p, pl.len.size() * 4, hipMemcpyDeviceToHost, st));
    }
    if (label) HIP_TRY(hipMemcpyAsync(label, s->d_label.p, (size_t)s->n_reads * 4, hipMemcpyDeviceToHost, st));
    if (log_post)
        HIP_TRY(hipMemcpyAsync(log_post, s->d_post.p, (size_t)s->n_reads * s->post_stride * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t c = 0; c < s->n_chunks; c++) {
        const ChunkState &cs = pl.state[c];
        if (cs.status != 0) {
            pl.any_fail = 1;
            // a failed chunk has no clustering: its reads get label 0 and zero posteriors, not whatever the device buffers held
            // (round 6: found by comparing a sliced call with the same call in one piece)
            const ChunkMeta &cm = s->h_chunks[c];
            for (uint32_t r = 0; r < cm.n_reads; r++) {
                if (label) label[cm.read_first + r] = 0;
                if (log_post)
                    for (uint32_t t = 0; t < s->post_stride; t++) log_post[(size_t)(cm.read_first + r) * s->post_stride + t] = 0.0;
            }
        }
        if (result) {
            result[c].score = cs.status == 0 ? cs.score : 0.0;
            result[c].cluster_num = cs.status == 0 ? cs.k : 1;
            result[c].status = cs.status;
            result[c].polish_rounds = cs.rounds;
            result[c].n_variants = cs.dim;
        }
    }
    if (want_ops)
        for (uint32_t g = 0; g < s->n_reads; g++) pl.ops_total += pl.len[g];
    if (want_cons)
        for (uint32_t c = 0; c < s->n_chunks; c++) pl.cons_total += pl.len[s->n_reads + c];
    return 0;
}

// cons_out / ops_out: the caller's arrays; cons_off / ops_out_off: the entries of THIS session's chunks / reads (n_chunks + 1 and
// n_reads + 1 of them are written); cons_base / ops_base: where this session's bytes start inside the arrays
int fetch_finish(jtk_lc_session_t *s, FetchPlan &pl, uint8_t *cons_out, uint64_t *cons_off, uint64_t cons_base, uint64_t cons_cap,
                 uint8_t *ops_out, uint64_t *ops_out_off, uint64_t ops_base, uint64_t ops_cap) {
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = s->stream;
    if (pl.want_cons && cons_base + pl.cons_total > cons_cap) return fail(JTK_ERR_INVALID_ARG, "cons_cap too small");
    if (pl.want_ops && ops_base + pl.ops_total > ops_cap) return fail(JTK_ERR_INVALID_ARG, "ops_cap too small");
    std::vector<uint64_t> off;   // (source of an asynchronous upload: lives until the stream has been waited for, below)
    if (pl.want_cons || pl.want_ops) {
        // offsets local to the session (the device packs from 0), written to the caller's arrays with the base added
        off.resize((size_t)s->n_reads + s->n_chunks + 2);
        uint64_t *ooff = off.data(), *coff = off.data() + s->n_reads + 1;
        uint64_t oo = 0, co = 0;
        for (uint32_t g = 0; g < s->n_reads; g++) {
            ooff[g] = oo;
            oo += pl.want_ops ? pl.len[g] : 0;
        }
        ooff[s->n_reads] = oo;
        for (uint32_t c = 0; c < s->n_chunks; c++) {
            coff[c] = co;
            co += pl.want_cons ? pl.len[s->n_reads + c] : 0;
        }
        coff[s->n_chunks] = co;
        int rc;
        if (pl.want_ops && !s->d_out_ops.p && (rc = dev_alloc<uint8_t>(s->d_out_ops, s->ops_bytes + 8))) return rc;
        if (pl.want_cons && !s->d_out_cons.p && (rc = dev_alloc<uint8_t>(s->d_out_cons, s->tmpl_bytes + 8))) return rc;
        HIP_TRY(hipMemcpyAsync(s->d_out_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, st));
        launch_gather(st, s->n_reads, s->n_chunks, s->d_reads.as<ReadMeta>(), s->d_chunks.as<ChunkMeta>(), s->d_state.as<ChunkState>(),
                      s->bufs, s->d_out_off.as<uint64_t>(), s->d_out_off.as<uint64_t>() + s->n_reads + 1,
                      pl.want_ops ? s->d_out_ops.as<uint8_t>() : nullptr, pl.want_cons ? s->d_out_cons.as<uint8_t>() : nullptr);
        if (pl.want_ops && oo) HIP_TRY(hipMemcpyAsync(ops_out + ops_base, s->d_out_ops.p, oo, hipMemcpyDeviceToHost, st));
        if (pl.want_cons && co) HIP_TRY(hipMemcpyAsync(cons_out + cons_base, s->d_out_cons.p, co, hipMemcpyDeviceToHost, st));
        if (pl.want_ops)
            for (uint32_t g = 0; g <= s->n_reads; g++) ops_out_off[g] = ops_base + ooff[g];
        if (pl.want_cons)
            for (uint32_t c = 0; c <= s->n_chunks; c++) cons_off[c] = cons_base + coff[c];
    }
    hipEvent_t ev1;
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    (void)hipEventElapsedTime(&ms, pl.ev0, ev1);
    g_timing.d2h_ms = ms;
    (void)hipEventDestroy(pl.ev0);
    (void)hipEventDestroy(ev1);
    pl.ev0 = nullptr;
    return 0;
}

// chunks that went through clustering_recursive's split: the merged clustering replaces the first pass's
int fetch_split_results(jtk_lc_session_t *s, uint32_t *label, double *log_post, jtk_lc_result_t *result) {
    int any_fail = 0;
    for (uint32_t c = 0; c < s->n_chunks && c < s->split.size(); c++) {
        const SplitResult &sr = s->split[c];
        if (sr.k == 0) continue;
        const ChunkMeta &cm = s->h_chunks[c];
        if (sr.status != 0) any_fail = 1;
        if (result) {
            result[c].score = sr.status == 0 ? sr.score : 0.0;
            result[c].cluster_num = sr.status == 0 ? sr.k : 1;
            result[c].status = sr.status;
        }
        for (uint32_t r = 0; r < cm.n_reads; r++) {
            const uint32_t g = cm.read_first + r;
            if (label) label[g] = sr.status == 0 ? sr.asn[r] : 0;
            if (log_post)
                for (uint32_t t = 0; t < s->post_stride; t++)
                    log_post[(size_t)g * s->post_stride + t] = (sr.status == 0 && t < sr.k) ? sr.post[(size_t)r * sr.k + t] : 0.0;
        }
    }
    return any_fail;
}
}  // namespace

int jtk_lc_session_fetch(jtk_lc_session_t *s, uint32_t *label, double *log_post, jtk_lc_result_t *result,
                         uint8_t *cons_out, uint64_t *cons_off, uint64_t cons_cap, uint8_t *ops_out,
                         uint64_t *ops_out_off, uint64_t ops_cap) {
    g_last_error.clear();
    if (!s) return fail(JTK_ERR_INVALID_ARG, "null session");
    FetchPlan pl;
    int rc = fetch_begin(s, pl, label, log_post, result, cons_out && cons_off, ops_out && ops_out_off);
    if (rc == 0) rc = fetch_finish(s, pl, cons_out, cons_off, 0, cons_cap, ops_out, ops_out_off, 0, ops_cap);
    if (pl.ev0) (void)hipEventDestroy(pl.ev0);
    if (rc) return rc;
    const int any_fail = pl.any_fail | fetch_split_results(s, label, log_post, result);
    return any_fail ? fail(JTK_ERR_CHUNK_FAILED, "at least one chunk failed; see result[].status") : 0;
}


// ---- clustering_recursive's split branch (mod.rs:138-189), driven from the host --------------------------
namespace {

const uint32_t UPPER_COPY_NUM = JTK_MAX_COPY + 1;  // mod.rs:85
const uint32_t BRANCH_NUM = 4;                     // mod.rs:139

// One clustering_recursive call of a chunk, waiting for its clustering() or for its sub-calls.
struct SplitFrame {
    std::vector<uint8_t> tmpl;              // bases the call starts from (a sub-call polishes them first, mod.rs:153-155)
    std::vector<uint32_t> rid;              // its reads: indices into the session's batch
    std::vector<std::vector<uint8_t>> ops;  // their ops against tmpl
    uint32_t copy_num = 0;
    bool have = false;                      // the device pass of this call has run:
    SplitResult own;                        //   clustering() with min(copy_num, BRANCH_NUM-or-itself) clusters
    std::vector<uint8_t> cons;              //   the consensus it clustered on
    std::vector<std::vector<uint8_t>> cops; //   and the ops re-threaded onto it
    std::vector<uint32_t> copy_numbers;     // estim_copy_num of the split
    std::vector<SplitResult> kids;          // finished sub-calls, in cluster order
};
struct SplitChunk {
    uint32_t chunk = 0;
    uint64_t rng[4];
    std::vector<SplitFrame> stack;
    bool done = false;
};

void seed_from_u64(uint64_t seed, uint64_t out[4]) {  // rand_core SeedableRng::seed_from_u64 for a 32-byte seed: SplitMix64
    for (int i = 0; i < 4; i++) {
        seed += 0x9e3779b97f4a7c15ULL;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        out[i] = z ^ (z >> 31);
    }
}
void rng_skip(uint64_t s[4], uint64_t draws) {  // the xoshiro256 state after `draws` outputs
    for (uint64_t i = 0; i < draws; i++) {
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = (s[3] << 45) | (s[3] >> 19);
    }
}

// estim_copy_num (mod.rs:223-243): one copy per cluster, every further copy to the cluster whose read count is
// farthest from coverage * copies (max_by keeps the last maximum)
std::vector<uint32_t> estim_copy_num(const std::vector<uint32_t> &asn, uint32_t k, uint32_t copy_num, double coverage) {
    std::vector<double> counts(k, 0.0);
    for (uint32_t a : asn) counts[a] += 1.0;
    std::vector<uint32_t> cp(k, 1);
    for (uint32_t it = k; it < copy_num; it++) {
        uint32_t arg = 0;
        double best = 0.0;
        for (uint32_t c = 0; c < k; c++) {
            const double d = counts[c] - coverage * (double)cp[c], v = d * d;
            if (c == 0 || !(v < best)) {
                best = v;
                arg = c;
            }
        }
        cp[arg] += 1;
    }
    return cp;
}

// the merge of mod.rs:161-187
SplitResult merge_split(const SplitFrame &f) {
    const uint32_t k = f.own.k;
    std::vector<uint32_t> offsets(k), pointers(k, 0);
    uint32_t total = 0;
    for (uint32_t c = 0; c < k; c++) {
        offsets[c] = total;
        total += f.kids[c].k;
    }
    SplitResult r;
    r.k = total;
    double sub = 0.0;
    for (uint32_t c = 0; c < k; c++) sub += f.kids[c].score;
    r.score = sub + f.own.score;
    const size_t n = f.own.asn.size();
    r.asn.resize(n);
    r.post.resize(n * total);
    for (size_t i = 0; i < n; i++) {
        const uint32_t a = f.own.asn[i], pt = pointers[a]++;
        const SplitResult &kid = f.kids[a];
        double *po = &r.post[i * total];
        uint32_t w = 0;
        for (uint32_t c = 0; c < k; c++) {
            const double lk = f.own.post[i * k + c] - jtk_log((double)f.kids[c].k);
            for (uint32_t t = 0; t < f.kids[c].k; t++) po[w++] = lk;
        }
        for (uint32_t t = 0; t < kid.k; t++) po[t + offsets[a]] += kid.post[(size_t)pt * kid.k + t] + jtk_log((double)kid.k);
        double sum = 0.0;
        for (uint32_t t = 0; t < total; t++) sum += jtk_exp(po[t]);
        if (!(std::fabs(1.0 - sum) < 0.0001)) r.status = JTK_ERR_CHUNK_FAILED;  // the reference asserts (mod.rs:184)
        r.asn[i] = offsets[a] + kid.asn[pt];
    }
    return r;
}

}  // namespace

static int run_split(jtk_lc_session_t *s) {
    jtk_lc_timing_t acc = g_timing;
    // ---- what the batch pass left for the split chunks: labels, posteriors, consensus, ops, draws
    std::vector<uint32_t> label(s->n_reads);
    std::vector<double> post((size_t)s->n_reads * s->post_stride);
    std::vector<jtk_lc_result_t> res(s->n_chunks);
    std::vector<uint8_t> cons(s->tmpl_bytes + 8), ops_out(s->ops_bytes + 8);
    std::vector<uint64_t> cons_off(s->n_chunks + 1), ops_off(s->n_reads + 1);
    std::vector<ChunkState> state(s->n_chunks);
    s->split.clear();  // fetch below must see the batch pass's own results
    int rc = jtk_lc_session_fetch(s, label.data(), post.data(), res.data(), cons.data(), cons_off.data(), cons.size(),
                                  ops_out.data(), ops_off.data(), ops_out.size());
    if (rc != 0 && rc != JTK_ERR_CHUNK_FAILED) return rc;
    HIP_TRY(hipMemcpy(state.data(), s->d_state.p, state.size() * sizeof(ChunkState), hipMemcpyDeviceToHost));
    acc.d2h_ms = 0;
    s->split.assign(s->n_chunks, SplitResult());
    std::vector<SplitChunk> work;
    for (uint32_t c = 0; c < s->n_chunks; c++) {
        if (s->h_copy0[c] < UPPER_COPY_NUM) continue;
        const ChunkMeta &cm = s->h_chunks[c];
        if (res[c].status != 0) {
            s->split[c].k = 1;
            s->split[c].status = res[c].status;
            continue;
        }
        if (res[c].cluster_num > s->post_stride) {  // (unreachable since session_create checks copy_num; per chunk anyway)
            s->split[c].k = 1;
            s->split[c].status = JTK_ERR_INVALID_ARG;
            continue;
        }
        SplitChunk w;
        w.chunk = c;
        seed_from_u64(cm.chunk_id * 3490ULL, w.rng);  // mod.rs:97
        rng_skip(w.rng, state[c].draws);
        SplitFrame f;
        f.copy_num = s->h_copy0[c];
        f.have = true;
        f.own.k = res[c].cluster_num;
        f.own.score = res[c].score;
        f.cons.assign(cons.begin() + cons_off[c], cons.begin() + cons_off[c + 1]);
        for (uint32_t r = 0; r < cm.n_reads; r++) {
            const uint32_t g = cm.read_first + r;
            f.rid.push_back(g);
            f.own.asn.push_back(label[g]);
            for (uint32_t t = 0; t < f.own.k; t++) f.own.post.push_back(post[(size_t)g * s->post_stride + t]);
            f.cops.emplace_back(ops_out.begin() + ops_off[g], ops_out.begin() + ops_off[g + 1]);
        }
        w.stack.push_back(std::move(f));
        work.push_back(std::move(w));
    }
    // ---- rounds: advance every chunk to its next clustering() call, run those calls as one resident batch
    for (;;) {
        std::vector<SplitChunk *> waiting;
        for (SplitChunk &w : work) {
            while (!w.done) {
                SplitFrame &f = w.stack.back();
                if (!f.have) break;
                SplitResult out;
                bool finished = false;
                if (f.own.status != 0) {
                    out.k = 1;
                    out.status = f.own.status;
                    w.stack.resize(1);  // the chunk fails as a whole
                    finished = true;
                } else if (f.copy_num < UPPER_COPY_NUM || f.own.k <= 1) {  // mod.rs:136-137, :146-148
                    out = std::move(f.own);
                    finished = true;
                } else if (f.kids.size() == f.own.k) {
                    out = merge_split(f);
                    finished = true;
                } else {
                    if (f.copy_numbers.empty())
                        f.copy_numbers = estim_copy_num(f.own.asn, f.own.k, f.copy_num, s->params.haploid_coverage);
                    const uint32_t c = (uint32_t)f.kids.size(), cp = f.copy_numbers[c];
                    SplitFrame kid;  // filter_sub_clusters (mod.rs:198-221)
                    for (size_t i = 0; i < f.rid.size(); i++)
                        if (f.own.asn[i] == c) {
                            kid.rid.push_back(f.rid[i]);
                            kid.ops.push_back(f.cops[i]);
                        }
                    if (cp < 2 || kid.rid.empty()) {
                        // clustering() returns at once, with no draw (pseudo_mcmc.rs:86-88): the polish before it
                        // (mod.rs:153-155) cannot reach the result and is not run
                        SplitResult t;
                        t.k = 1;
                        t.asn.assign(kid.rid.size(), 0);
                        t.post.assign(kid.rid.size(), 0.0);
                        f.kids.push_back(std::move(t));
                        continue;
                    }
                    kid.tmpl = f.cons;
                    kid.copy_num = cp;
                    w.stack.push_back(std::move(kid));
                    continue;
                }
                if (finished) {
                    w.stack.pop_back();
                    if (w.stack.empty()) {
                        // the merged clustering has up to copy_num clusters: the caller's rows must hold them
                        if (out.status == 0 && out.k > s->post_stride) out.status = JTK_ERR_INVALID_ARG;
                        s->split[w.chunk] = std::move(out);
                        w.done = true;
                    } else if (out.status != 0) {
                        w.stack.back().own.status = out.status;
                    } else {
                        w.stack.back().kids.push_back(std::move(out));
                    }
                }
            }
            if (!w.done) waiting.push_back(&w);
        }
        if (waiting.empty()) break;
        // pack the waiting calls
        const size_t nb = waiting.size();
        std::vector<jtk_lc_chunk_t> chunks(nb);
        std::vector<ChunkExtra> extra(nb);
        std::vector<uint8_t> tmpl, reads, opsv, strand;
        std::vector<uint64_t> roff(1, 0), ooff(1, 0);
        for (size_t b = 0; b < nb; b++) {
            const SplitChunk &w = *waiting[b];
            const SplitFrame &f = w.stack.back();
            const ChunkMeta &cm = s->h_chunks[w.chunk];
            chunks[b].chunk_id = cm.chunk_id;
            chunks[b].copy_num = f.copy_num;
            chunks[b].n_reads = (uint32_t)f.rid.size();
            chunks[b].tmpl_off = tmpl.size();
            chunks[b].tmpl_len = f.tmpl.size();
            chunks[b].read_first = strand.size();
            extra[b].radius = cm.radius;
            extra[b].local_coverage = cm.local_coverage;
            memcpy(extra[b].rng, w.rng, 32);
            extra[b].take_num = 0;
            tmpl.insert(tmpl.end(), f.tmpl.begin(), f.tmpl.end());
            for (size_t i = 0; i < f.rid.size(); i++) {
                const uint32_t g = f.rid[i];
                reads.insert(reads.end(), s->h_read_bases.begin() + s->h_read_off[g], s->h_read_bases.begin() + s->h_read_off[g + 1]);
                roff.push_back(reads.size());
                opsv.insert(opsv.end(), f.ops[i].begin(), f.ops[i].end());
                ooff.push_back(opsv.size());
                strand.push_back(s->h_strand[g]);
            }
        }
        const uint32_t stride = JTK_MAX_COPY;
        jtk_lc_session_t *sub = nullptr;
        rc = session_create_ex(&s->params, nb, chunks.data(), tmpl.data(), reads.data(), roff.data(), opsv.data(), ooff.data(),
                               strand.data(), stride, s->device, extra.data(), 0 /* mod.rs:153 */, &sub);
        if (rc) return rc;
        std::unique_ptr<jtk_lc_session> guard(sub);
        if ((rc = run_batch(sub, 0))) return rc;
        for (int k = 0; k < JTK_K_COUNT; k++) {
            acc.kernel_ms[k] += g_timing.kernel_ms[k];
            acc.kernel_launches[k] += g_timing.kernel_launches[k];
        }
        acc.total_ms += g_timing.total_ms;
        const uint32_t nr = (uint32_t)strand.size();
        std::vector<uint32_t> lab(nr);
        std::vector<double> pst((size_t)nr * stride);
        std::vector<jtk_lc_result_t> rs(nb);
        std::vector<uint8_t> cs(sub->tmpl_bytes + 8), os(sub->ops_bytes + 8);
        std::vector<uint64_t> coff(nb + 1), ofo(nr + 1);
        std::vector<ChunkState> sst(nb);
        rc = jtk_lc_session_fetch(sub, lab.data(), pst.data(), rs.data(), cs.data(), coff.data(), cs.size(), os.data(),
                                  ofo.data(), os.size());
        if (rc != 0 && rc != JTK_ERR_CHUNK_FAILED) return rc;
        HIP_TRY(hipMemcpy(sst.data(), sub->d_state.p, sst.size() * sizeof(ChunkState), hipMemcpyDeviceToHost));
        for (size_t b = 0; b < nb; b++) {
            SplitChunk &w = *waiting[b];
            SplitFrame &f = w.stack.back();
            f.have = true;
            f.own.status = rs[b].status;
            if (rs[b].status != 0) continue;
            rng_skip(w.rng, sst[b].draws);
            f.own.k = rs[b].cluster_num;
            f.own.score = rs[b].score;
            f.cons.assign(cs.begin() + coff[b], cs.begin() + coff[b + 1]);
            const uint32_t first = (uint32_t)chunks[b].read_first;
            for (uint32_t r = 0; r < chunks[b].n_reads; r++) {
                f.own.asn.push_back(lab[first + r]);
                for (uint32_t t = 0; t < f.own.k; t++) f.own.post.push_back(pst[(size_t)(first + r) * stride + t]);
                f.cops.emplace_back(os.begin() + ofo[first + r], os.begin() + ofo[first + r + 1]);
            }
        }
    }
    g_timing = acc;
    return 0;
}

int jtk_lc_session_destroy(jtk_lc_session_t *s) {
    if (!s) return 0;
    (void)hipSetDevice(s->device);
    delete s;
    return 0;
}

static int run_slice(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                     const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off, const uint8_t *ops,
                     const uint64_t *ops_off, const uint8_t *strand, int skip_polish, uint32_t *label, double *log_post,
                     uint32_t post_stride, jtk_lc_result_t *result, uint8_t *cons_out, uint64_t *cons_off,
                     uint64_t cons_cap, uint8_t *ops_out, uint64_t *ops_out_off, uint64_t ops_cap, int device) {
    jtk_lc_session_t *s = nullptr;
    int rc = jtk_lc_session_create(params, n_chunks, chunks, tmpl_bases, read_bases, read_off, ops, ops_off, strand,
                                   post_stride, device, &s);
    if (rc) return rc;
    rc = jtk_lc_session_run(s, skip_polish);
    if (rc == 0)
        rc = jtk_lc_session_fetch(s, label, log_post, result, cons_out, cons_off, cons_cap, ops_out, ops_out_off, ops_cap);
    const std::string keep = g_last_error;
    jtk_lc_session_destroy(s);
    g_last_error = keep;
    return rc;
}

// The one-shot entry points on a large batch: up to four slices of the batch run as independent sessions on their own streams
// and host threads, so that one slice's pair-HMM passes fill the CUs another slice's chain kernel leaves idle during
// its tail (the overlap bench.py gets from four resident batches, §6 of DESIGN.md).  Chunks are independent (the RNG
// is seeded per chunk), so the results do not depend on the slicing.  A slice keeps >= 500 chunks: below that the
// tail of its own chain kernel is all there is to hide.  JTK_LC_SLICES overrides the count (tests, tuning).
static int run_once(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                    const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off, const uint8_t *ops,
                    const uint64_t *ops_off, const uint8_t *strand, int skip_polish, uint32_t *label, double *log_post,
                    uint32_t post_stride, jtk_lc_result_t *result, uint8_t *cons_out, uint64_t *cons_off,
                    uint64_t cons_cap, uint8_t *ops_out, uint64_t *ops_out_off, uint64_t ops_cap, const int *devices,
                    size_t n_devices) {
    // several devices: the same slicing, consecutive slices dealt to consecutive devices (a device's slices overlap
    // each other as on one GPU; devices share nothing)
    if (!devices || n_devices == 0) return fail(JTK_ERR_INVALID_ARG, "no device given");
    size_t per_dev = std::min<size_t>(4, n_chunks / n_devices / 500);  // 2500 chunks: 2.08 / 1.80 / 2.08 / 2.17 s in 3 / 4 / 5 / 6 slices
    if (const char *e = getenv("JTK_LC_SLICES")) per_dev = (size_t)atoi(e);
    if (per_dev < 1) per_dev = 1;
    // per_dev slices of a device run side by side.  A batch whose workspaces do not fit beside each other that way (deep
    // pile-ups: 296 KB of row sums / tables per read of a 2 kbp chunk) is cut into MORE slices, which the per_dev worker threads
    // of the device take one after the other: the workspace in use stays bounded by what per_dev slices need, and the
    // blocks a finished slice returns to the pool are what the next one takes.
    size_t slices_per_dev = per_dev;
    if (chunks && n_chunks) {
        uint64_t est = 0, max_len = 0, max_rd = 0;
        for (size_t c = 0; c < n_chunks; c++) {
            const uint64_t cap = chunks[c].tmpl_len + chunks[c].tmpl_len / 8 + 64;
            est += (uint64_t)chunks[c].n_reads * (cap + 1) * (JTK_ACC_N * 8 + 16);  // row sums / tables + ops / deltas
            max_len = std::max<uint64_t>(max_len, cap);
            if (read_off) {
                const uint64_t r0 = chunks[c].read_first, r1 = r0 + chunks[c].n_reads;
                if (r1 > r0) max_rd = std::max<uint64_t>(max_rd, (read_off[r1] - read_off[r0]) / (r1 - r0) + 64);
            }
        }
        size_t free_b = 0, total_b = 0;
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (hipSetDevice(devices[0]) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) {
            const uint64_t stripes = 4096ull * (max_len + max_rd + 32) * 64 * 16;   // the device's shared pair-HMM stripe set
            const double budget = (0.80 * (double)total_b - (double)stripes) / (double)per_dev;  // pooled blocks count as free
            if (budget > 0) {
                const size_t need = (size_t)std::ceil((double)est / (double)n_devices / budget);
                if (need > slices_per_dev) slices_per_dev = need;
            }
        }
        (void)hipSetDevice(cur);
    }
    size_t n_slices = slices_per_dev * n_devices;
    if (n_slices > n_chunks) n_slices = n_chunks;
    if (n_slices < 2 || !params || !chunks || !read_off || !ops_off || !label || !log_post || !result)
        return run_slice(params, n_chunks, chunks, tmpl_bases, read_bases, read_off, ops, ops_off, strand, skip_polish, label,
                         log_post, post_stride, result, cons_out, cons_off, cons_cap, ops_out, ops_out_off, ops_cap, devices[0]);
    g_last_error.clear();
    uint64_t n_reads = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        if (chunks[c].read_first != n_reads) return fail(JTK_ERR_INVALID_ARG, "chunks must list their reads contiguously in order");
        n_reads += chunks[c].n_reads;
    }
    // slice boundaries: equal shares of the reads
    std::vector<size_t> first(n_slices + 1, n_chunks);
    first[0] = 0;
    {
        size_t sl = 1;
        uint64_t seen = 0;
        for (size_t c = 0; c < n_chunks && sl < n_slices; c++) {
            seen += chunks[c].n_reads;
            if (seen * n_slices >= n_reads * sl) first[sl++] = c + 1;
        }
    }
    const bool want_cons = cons_out && cons_off, want_ops = ops_out && ops_out_off;
    // Every slice writes its results straight into the caller's arrays.  Where a slice's consensus / ops start depends on the
    // lengths of the slices before it: a slice learns its own lengths (fetch_begin), waits for its predecessor to publish where it
    // ends, publishes its own end and only then packs and copies (fetch_finish) -- no per-slice staging vectors, no stitching pass.
    struct Slice {
        std::vector<jtk_lc_chunk_t> chunks;
        int rc = 0;
        std::string error;
        jtk_lc_timing_t timing;
    };
    std::vector<Slice> slices(n_slices);
    std::vector<std::function<void()>> jobs(n_slices);
    std::mutex base_mutex;
    std::condition_variable base_cv;
    std::vector<uint64_t> cons_base(n_slices + 1, 0), ops_base(n_slices + 1, 0);
    std::vector<char> base_known(n_slices + 1, 0);
    base_known[0] = 1;
    for (size_t sl = 0; sl < n_slices; sl++) {
        Slice &S = slices[sl];
        const size_t c0 = first[sl], c1 = first[sl + 1];
        memset(&S.timing, 0, sizeof S.timing);
        const uint64_t r0 = c0 < c1 ? chunks[c0].read_first : n_reads;
        const uint64_t r1 = c0 < c1 ? chunks[c1 - 1].read_first + chunks[c1 - 1].n_reads : n_reads;
        if (c0 < c1) {
            S.chunks.assign(chunks + c0, chunks + c1);
            for (auto &ch : S.chunks) ch.read_first -= r0;
        }
        const int device = devices[std::min(sl / slices_per_dev, n_devices - 1)];
        jobs[sl] = ([=, &S, &base_mutex, &base_cv, &cons_base, &ops_base, &base_known]() {
            jtk_lc_session_t *s = nullptr;
            FetchPlan pl;
            if (c0 < c1) {
                S.rc = jtk_lc_session_create(params, c1 - c0, S.chunks.data(), tmpl_bases, read_bases, read_off + r0, ops, ops_off + r0,
                                             strand + r0, post_stride, device, &s);
                if (S.rc == 0) S.rc = jtk_lc_session_run(s, skip_polish);
                if (S.rc == 0)
                    S.rc = fetch_begin(s, pl, label + r0, log_post + r0 * post_stride, result + c0, want_cons, want_ops);
            }
            const bool ok = c0 < c1 && S.rc == 0;
            uint64_t cb = 0, ob = 0;
            {   // (also on failure: the slices behind this one are waiting for its end)
                std::unique_lock<std::mutex> lock(base_mutex);
                base_cv.wait(lock, [&]() { return base_known[sl] != 0; });
                cb = cons_base[sl];
                ob = ops_base[sl];
                cons_base[sl + 1] = cb + (ok ? pl.cons_total : 0);
                ops_base[sl + 1] = ob + (ok ? pl.ops_total : 0);
                base_known[sl + 1] = 1;
            }
            base_cv.notify_all();
            if (ok) {
                S.rc = fetch_finish(s, pl, cons_out, want_cons ? cons_off + c0 : nullptr, cb, cons_cap, ops_out,
                                    want_ops ? ops_out_off + r0 : nullptr, ob, ops_cap);
                if (S.rc == 0 && (pl.any_fail | fetch_split_results(s, label + r0, log_post + r0 * post_stride, result + c0)))
                    S.rc = fail(JTK_ERR_CHUNK_FAILED, "at least one chunk failed; see result[].status");
            } else {   // nothing from this slice: its chunks and reads get empty ranges
                if (want_cons)
                    for (size_t c = c0; c <= c1 && c <= n_chunks; c++) cons_off[c] = cb;
                if (want_ops)
                    for (uint64_t g = r0; g <= r1; g++) ops_out_off[g] = ob;
            }
            if (pl.ev0) (void)hipEventDestroy(pl.ev0);
            S.error = g_last_error;   // thread-local in the slice's thread
            S.timing = g_timing;
            if (s) jtk_lc_session_destroy(s);
        });
    }
    {   // per device: per_dev worker threads take the device's slices in order (a slice only ever waits for an EARLIER slice's
        // lengths, and those are taken first: no cycle)
        std::vector<std::thread> threads;
        std::vector<std::atomic<size_t>> next(n_devices);
        for (size_t d = 0; d < n_devices; d++) next[d] = d * slices_per_dev;
        for (size_t d = 0; d < n_devices; d++) {
            const size_t end = std::min(n_slices, (d + 1) * slices_per_dev);
            for (size_t w = 0; w < per_dev && d * slices_per_dev + w < end; w++)
                threads.emplace_back([&, d, end]() {
                    for (size_t sl = next[d].fetch_add(1); sl < end; sl = next[d].fetch_add(1))
                        if (jobs[sl]) jobs[sl]();
                });
        }
        for (auto &t : threads) t.join();
    }
    int rc = 0;
    memset(&g_timing, 0, sizeof g_timing);
    for (size_t sl = 0; sl < n_slices; sl++) {
        Slice &S = slices[sl];
        if (first[sl] >= first[sl + 1]) continue;
        if (S.rc != 0 && (rc == 0 || rc == JTK_ERR_CHUNK_FAILED)) {
            rc = S.rc;
            g_last_error = S.error;
        }
        g_timing.h2d_ms += S.timing.h2d_ms;
        g_timing.d2h_ms += S.timing.d2h_ms;
        g_timing.total_ms = std::max(g_timing.total_ms, S.timing.total_ms);   // the slices run side by side
        for (int j = 0; j < 2; j++) g_timing.chain_lds_bytes[j] = std::max(g_timing.chain_lds_bytes[j], S.timing.chain_lds_bytes[j]);
        for (int k = 0; k < JTK_K_COUNT; k++) {
            g_timing.kernel_ms[k] += S.timing.kernel_ms[k];
            g_timing.kernel_launches[k] += S.timing.kernel_launches[k];
        }
    }
    return rc;
}

int jtk_lc_cluster_chunks(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                          const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                          const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, uint32_t *label,
                          double *log_post, uint32_t post_stride, jtk_lc_result_t *result, uint8_t *cons_out,
                          uint64_t *cons_off, uint64_t cons_cap, uint8_t *ops_out, uint64_t *ops_out_off,
                          uint64_t ops_cap, int device) {
    return run_once(params, n_chunks, chunks, tmpl_bases, read_bases, read_off, ops, ops_off, strand, 0, label,
                    log_post, post_stride, result, cons_out, cons_off, cons_cap, ops_out, ops_out_off, ops_cap, &device, 1);
}

// The same stage call over several GPUs of one node from ONE host process.  The chunks are dealt to the listed devices by
// longest-processing-time-first over the cost model of jtk_amd/sharding.py (`chunk_cost`: pair-HMM cells of the polishing
// passes + Metropolis steps per candidate k), the partition `bench.py --gpus N` uses between ranks: a device's share is in
// general NOT a contiguous range, so it is gathered into its own flat batch (templates stay where they are: chunks carry
// offsets), run as on a single device (sliced and overlapped), and its results are scattered back to the caller's order.
// The path has no exchange step, so there is no collective: this is SURVEY 8(b)'s `device_mask` as an explicit list.
static double chunk_cost(const jtk_lc_chunk_t &c) {
    const double n_k = (double)std::max<uint32_t>(1, std::min<uint32_t>(c.copy_num, 7) - (c.copy_num ? 1 : 0));
    return (double)c.n_reads * (double)c.tmpl_len * 3 * 61 * 2 + 20.0 * 2000.0 * (double)c.n_reads * n_k * 40.0;
}

int jtk_lc_cluster_chunks_multi(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                                const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                                const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, uint32_t *label,
                                double *log_post, uint32_t post_stride, jtk_lc_result_t *result, uint8_t *cons_out,
                                uint64_t *cons_off, uint64_t cons_cap, uint8_t *ops_out, uint64_t *ops_out_off,
                                uint64_t ops_cap, const int *devices, size_t n_devices) {
    g_last_error.clear();
    if (!devices || n_devices == 0) return fail(JTK_ERR_INVALID_ARG, "no device given");
    if (n_devices == 1 || n_chunks < 2 * n_devices || !params || !chunks || !read_bases || !read_off || !ops || !ops_off ||
        !strand || !label || !log_post || !result)  // one device, a tiny batch, or arguments the single-device path reports on
        return run_once(params, n_chunks, chunks, tmpl_bases, read_bases, read_off, ops, ops_off, strand, 0, label,
                        log_post, post_stride, result, cons_out, cons_off, cons_cap, ops_out, ops_out_off, ops_cap, devices, 1);
    uint64_t n_reads = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        if (chunks[c].read_first != n_reads) return fail(JTK_ERR_INVALID_ARG, "chunks must list their reads contiguously in order");
        n_reads += chunks[c].n_reads;
    }
    // LPT: costliest chunk first (ties: input order), each to the device with the least load so far (ties: first listed)
    std::vector<size_t> by_cost(n_chunks);
    std::vector<double> cost(n_chunks);
    for (size_t c = 0; c < n_chunks; c++) {
        by_cost[c] = c;
        cost[c] = chunk_cost(chunks[c]);
    }
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
    std::vector<double> load(n_devices, 0.0);
    std::vector<std::vector<size_t>> share(n_devices);
    for (size_t c : by_cost) {
        const size_t d = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
        share[d].push_back(c);
        load[d] += cost[c];
    }
    const bool want_cons = cons_out && cons_off, want_ops = ops_out && ops_out_off;
    struct Share {
        std::vector<jtk_lc_chunk_t> chunks;
        std::vector<uint8_t> read_bases, ops, strand, cons, ops_o;
        std::vector<uint64_t> read_off, ops_off, cons_off, ops_o_off;
        std::vector<uint32_t> label;
        std::vector<double> post;
        std::vector<jtk_lc_result_t> result;
        int rc = 0;
        std::string error;
        jtk_lc_timing_t timing;
    };
    std::vector<Share> shares(n_devices);
    std::vector<std::thread> threads;
    for (size_t d = 0; d < n_devices; d++) {
        std::sort(share[d].begin(), share[d].end());  // a device's chunks keep the caller's order
        threads.emplace_back([&, d]() {
            Share &S = shares[d];
            const std::vector<size_t> &ids = share[d];
            memset(&S.timing, 0, sizeof S.timing);
            if (ids.empty()) return;
            uint64_t nr = 0, nb = 0, no = 0, cons_need = 64, ops_need = 64;
            for (size_t c : ids) {
                const uint64_t r0 = chunks[c].read_first, r1 = r0 + chunks[c].n_reads;
                nr += r1 - r0;
                nb += read_off[r1] - read_off[r0];
                no += ops_off[r1] - ops_off[r0];
                cons_need += chunks[c].tmpl_len + chunks[c].tmpl_len / 4 + 64;
                ops_need += (uint64_t)chunks[c].n_reads * (chunks[c].tmpl_len / 4 + 72);
            }
            ops_need += no;
            S.chunks.reserve(ids.size());
            S.read_bases.resize(nb ? nb : 1);
            S.ops.resize(no ? no : 1);
            S.strand.resize(nr ? nr : 1);
            S.read_off.resize(nr + 1);
            S.ops_off.resize(nr + 1);
            S.label.resize(nr ? nr : 1);
            S.post.resize(nr ? nr * (size_t)post_stride : 1);
            S.result.resize(ids.size());
            if (want_cons) {
                S.cons.resize(cons_need);
                S.cons_off.resize(ids.size() + 1);
            }
            if (want_ops) {
                S.ops_o.resize(ops_need);
                S.ops_o_off.resize(nr + 1);
            }
            uint64_t r = 0, b = 0, o = 0;
            for (size_t c : ids) {
                jtk_lc_chunk_t ch = chunks[c];
                const uint64_t r0 = ch.read_first, r1 = r0 + ch.n_reads;
                ch.read_first = r;
                S.chunks.push_back(ch);
                memcpy(S.read_bases.data() + b, read_bases + read_off[r0], read_off[r1] - read_off[r0]);
                memcpy(S.ops.data() + o, ops + ops_off[r0], ops_off[r1] - ops_off[r0]);
                memcpy(S.strand.data() + r, strand + r0, r1 - r0);
                for (uint64_t g = r0; g < r1; g++, r++) {
                    S.read_off[r] = b + (read_off[g] - read_off[r0]);
                    S.ops_off[r] = o + (ops_off[g] - ops_off[r0]);
                }
                b += read_off[r1] - read_off[r0];
                o += ops_off[r1] - ops_off[r0];
            }
            S.read_off[nr] = b;
            S.ops_off[nr] = o;
            S.rc = run_once(params, ids.size(), S.chunks.data(), tmpl_bases, S.read_bases.data(), S.read_off.data(), S.ops.data(),
                            S.ops_off.data(), S.strand.data(), 0, S.label.data(), S.post.data(), post_stride, S.result.data(),
                            want_cons ? S.cons.data() : nullptr, want_cons ? S.cons_off.data() : nullptr, S.cons.size(),
                            want_ops ? S.ops_o.data() : nullptr, want_ops ? S.ops_o_off.data() : nullptr, S.ops_o.size(),
                            &devices[d], 1);
            S.error = g_last_error;  // thread-local in the share's thread
            S.timing = g_timing;
        });
    }
    for (auto &t : threads) t.join();
    // scatter the shares back into the caller's order
    int rc = 0;
    memset(&g_timing, 0, sizeof g_timing);
    std::vector<uint32_t> dev_of(n_chunks), idx_of(n_chunks);
    for (size_t d = 0; d < n_devices; d++) {
        const Share &S = shares[d];
        if (S.rc != 0 && (rc == 0 || rc == JTK_ERR_CHUNK_FAILED)) {
            rc = S.rc;
            g_last_error = S.error;
        }
        g_timing.h2d_ms += S.timing.h2d_ms;
        g_timing.d2h_ms += S.timing.d2h_ms;
        g_timing.total_ms = std::max(g_timing.total_ms, S.timing.total_ms);  // the devices run side by side
        for (int j = 0; j < 2; j++) g_timing.chain_lds_bytes[j] = std::max(g_timing.chain_lds_bytes[j], S.timing.chain_lds_bytes[j]);
        for (int k = 0; k < JTK_K_COUNT; k++) {
            g_timing.kernel_ms[k] += S.timing.kernel_ms[k];
            g_timing.kernel_launches[k] += S.timing.kernel_launches[k];
        }
        for (size_t i = 0; i < share[d].size(); i++) {
            dev_of[share[d][i]] = (uint32_t)d;
            idx_of[share[d][i]] = (uint32_t)i;
        }
    }
    if (rc != 0 && rc != JTK_ERR_CHUNK_FAILED) return rc;
    uint64_t co = 0, oo = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        const Share &S = shares[dev_of[c]];
        const size_t i = idx_of[c];
        const uint64_t r0 = chunks[c].read_first, nr = chunks[c].n_reads, g0 = S.chunks[i].read_first;
        result[c] = S.result[i];
        memcpy(label + r0, S.label.data() + g0, nr * sizeof(uint32_t));
        memcpy(log_post + r0 * post_stride, S.post.data() + g0 * post_stride, nr * post_stride * sizeof(double));
        if (want_cons) {
            const uint64_t a = S.cons_off[i], len = S.cons_off[i + 1] - a;
            if (co + len > cons_cap) return fail(JTK_ERR_INVALID_ARG, "cons_cap too small");
            memcpy(cons_out + co, S.cons.data() + a, len);
            cons_off[c] = co;
            co += len;
        }
        if (want_ops) {
            const uint64_t a = S.ops_o_off[g0], len = S.ops_o_off[g0 + nr] - a;
            if (oo + len > ops_cap) return fail(JTK_ERR_INVALID_ARG, "ops_cap too small");
            memcpy(ops_out + oo, S.ops_o.data() + a, len);
            for (uint64_t g = 0; g < nr; g++) ops_out_off[r0 + g] = oo + (S.ops_o_off[g0 + g] - a);
            oo += len;
        }
    }
    if (want_cons) cons_off[n_chunks] = co;
    if (want_ops) ops_out_off[n_reads] = oo;
    return rc;
}

int jtk_lc_cluster_polished(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                            const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                            const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, uint32_t *label,
                            double *log_post, uint32_t post_stride, jtk_lc_result_t *result, int device) {
    return run_once(params, n_chunks, chunks, tmpl_bases, read_bases, read_off, ops, ops_off, strand, 1, label,
                    log_post, post_stride, result, nullptr, nullptr, 0, nullptr, nullptr, 0, &device, 1);
}

// kiley polish_until_converge_antidiagonal(template, seqs, ops, strands, HMMPolishConfig::new(radius, take_num, ignore_edge))
// for a batch of independent windows: what consensus::polish_seg (haplotyper/src/consensus/mod.rs:445-496, :476-483) and
// polish_segments.rs run on 2 kbp windows of contigs -- the same kernels as the stage's own polishing step.
int jtk_lc_polish_chunks(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                         const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                         const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, uint32_t radius,
                         uint32_t take_num, uint32_t ignore_edge, uint8_t *cons_out, uint64_t *cons_off, uint64_t cons_cap,
                         uint8_t *ops_out, uint64_t *ops_out_off, uint64_t ops_cap, jtk_lc_result_t *result, int device) {
    g_last_error.clear();
    if (!params) return fail(JTK_ERR_INVALID_ARG, "null params");
    if (n_chunks && (!chunks || !tmpl_bases || !read_bases || !read_off || !ops || !ops_off || !strand))
        return fail(JTK_ERR_INVALID_ARG, "null input");
    if (!cons_out || !cons_off || !ops_out || !ops_out_off) return fail(JTK_ERR_INVALID_ARG, "null output");
    std::vector<ChunkExtra> extra(n_chunks);
    std::vector<jtk_lc_chunk_t> ch(chunks, chunks + (chunks ? n_chunks : 0));
    for (size_t c = 0; c < n_chunks; c++) {
        memset(&extra[c], 0, sizeof extra[c]);
        // radius 0: derive it from the window length like the stage does (mod.rs:96,105)
        extra[c].radius = radius ? radius : (uint32_t)std::ceil((double)ch[c].tmpl_len * params->band_frac) / 2;
        extra[c].take_num = take_num;
        ch[c].copy_num = 1;  // no clustering happens; keeps every posterior row a single entry
    }
    jtk_lc_params_t pp = *params;
    if (pp.gains.max_homopolymer_len == 0) pp.gains.max_homopolymer_len = 1;  // polishing does not use the gains
    jtk_lc_session_t *s = nullptr;
    int rc = session_create_ex(&pp, n_chunks, ch.data(), tmpl_bases, read_bases, read_off, ops, ops_off, strand, 1, device,
                               extra.data(), ignore_edge, &s, true);
    if (rc) return rc;
    std::unique_ptr<jtk_lc_session> guard(s);
    s->resume_rng = false;
    if ((rc = run_batch(s, 0))) return rc;
    return jtk_lc_session_fetch(s, nullptr, nullptr, result, cons_out, cons_off, cons_cap, ops_out, ops_out_off, ops_cap);
}

// ---- the model refit of the stage preamble (model_tune.rs:96-156) ------------------------------------------------------
namespace {
const int FIT_COUNTS = 45;  // 9 transitions (M,I,D x M,I,D), mat_emit[16], ins_emit[20]

// M-step on summed counts: every row is divided by its sum; rows without mass keep the old values
void fit_mstep(const jtk_hmm_t &old, const double *cnt, jtk_hmm_t &out) {
    out = old;
    double *tr[3] = {&out.mat_mat, &out.ins_mat, &out.del_mat};
    for (int st = 0; st < 3; st++) {
        const double sum = (cnt[3 * st] + cnt[3 * st + 1]) + cnt[3 * st + 2];
        if (sum > 0.0)
            for (int q = 0; q < 3; q++) tr[st][q] = cnt[3 * st + q] / sum;
    }
    for (int x = 0; x < 4; x++) {
        const double *e = cnt + 9 + 4 * x;
        const double sum = ((e[0] + e[1]) + e[2]) + e[3];
        if (sum > 0.0)
            for (int q = 0; q < 4; q++) out.mat_emit[4 * x + q] = e[q] / sum;
    }
    for (int cx = 0; cx < 5; cx++) {
        const double *e = cnt + 25 + 4 * cx;
        const double sum = ((e[0] + e[1]) + e[2]) + e[3];
        if (sum > 0.0)
            for (int q = 0; q < 4; q++) out.ins_emit[4 * cx + q] = e[q] / sum;
    }
}
}  // namespace

// `rounds` x [ polish every training pile-up with HMMPolishConfig::new(band / 2, N, 0) (model_tune.rs:137-143), then one
// Baum-Welch step on all of them with the LARGEST band (fit_antidiagonal_par_multiple(&packs, bw / 2), :136,:144-151) ].
// Polishing and the expected counts run on the device; the M-step is a few dozen divisions on the host.
int jtk_lc_fit_model(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks, const uint8_t *tmpl_bases,
                     const uint8_t *read_bases, const uint64_t *read_off, const uint8_t *ops, const uint64_t *ops_off,
                     const uint8_t *strand, uint32_t rounds, jtk_hmm_t *forward_out, jtk_hmm_t *reverse_out, int device) {
    g_last_error.clear();
    if (!params || !forward_out || !reverse_out || !chunks || !read_off || !ops_off || !strand)
        return fail(JTK_ERR_INVALID_ARG, "null argument");
    size_t n_reads = 0, tmpl_total = 0;
    uint32_t max_bw = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        n_reads += chunks[c].n_reads;
        tmpl_total += (size_t)chunks[c].tmpl_len;
        max_bw = std::max<uint32_t>(max_bw, (uint32_t)std::ceil((double)chunks[c].tmpl_len * params->band_frac));
    }
    if (n_chunks == 0 || n_reads == 0) return fail(JTK_ERR_INVALID_ARG, "no training pile-up");  // assert!, model_tune.rs:135
    if (max_bw / 2 > JTK_WIDE_MAX_RADIUS) return fail(JTK_ERR_UNSUPPORTED, "band radius > 255");
    jtk_lc_params_t cur = *params;
    if (cur.gains.max_homopolymer_len == 0) cur.gains.max_homopolymer_len = 1;  // the gains play no part in the refit
    // working copies: consensus and ops change from round to round; band_width stays that of the unpolished chunk (:123)
    std::vector<jtk_lc_chunk_t> ch(chunks, chunks + n_chunks);
    std::vector<uint8_t> cons(2 * tmpl_total + 64 * n_chunks + 64), cops(2 * (size_t)ops_off[n_reads] + 64 * n_reads + 64);
    std::vector<uint64_t> coff(n_chunks + 1), ooff(ops_off, ops_off + n_reads + 1);
    std::vector<ChunkExtra> extra(n_chunks);
    memcpy(cops.data(), ops, (size_t)ops_off[n_reads]);
    uint64_t o = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        coff[c] = o;
        memcpy(cons.data() + o, tmpl_bases + chunks[c].tmpl_off, (size_t)chunks[c].tmpl_len);
        o += chunks[c].tmpl_len;
        memset(&extra[c], 0, sizeof extra[c]);
        extra[c].radius = (uint32_t)std::ceil((double)chunks[c].tmpl_len * params->band_frac) / 2;
        ch[c].copy_num = 1;
    }
    coff[n_chunks] = o;
    std::vector<uint8_t> cons2(cons.size()), cops2(cops.size());
    std::vector<uint64_t> coff2(n_chunks + 1), ooff2(n_reads + 1);
    std::vector<jtk_lc_result_t> res(n_chunks);
    std::vector<double> counts((size_t)n_reads * FIT_COUNTS), lks(n_reads);
    for (uint32_t round = 0; round < rounds; round++) {
        for (size_t c = 0; c < n_chunks; c++) {
            ch[c].tmpl_off = coff[c];
            ch[c].tmpl_len = coff[c + 1] - coff[c];
        }
        jtk_lc_session_t *s = nullptr;
        int rc = session_create_ex(&cur, n_chunks, ch.data(), cons.data(), read_bases, read_off, cops.data(), ooff.data(), strand,
                                   1, device, extra.data(), 0 /* ignore_edge, model_tune.rs:140 */, &s, true);
        if (rc) return rc;
        std::unique_ptr<jtk_lc_session> guard(s);
        s->resume_rng = false;
        if ((rc = run_batch(s, 0))) return rc;
        rc = jtk_lc_session_fetch(s, nullptr, nullptr, res.data(), cons2.data(), coff2.data(), cons2.size(), cops2.data(),
                                  ooff2.data(), cops2.size());
        if (rc) return rc;  // a training pile-up that fails fails the fit (the reference would panic)
        // ---- E-step on the polished pile-ups, every read with the largest band's radius
        {
            std::vector<ChunkMeta> wide(s->h_chunks);
            for (auto &cm : wide) {
                cm.radius = max_bw / 2;
                cm.take_num = 0;
            }
            // (checked before anything is allocated or queued: an early return must not hand a block with a pending memset
            // back to the pool)
            const size_t lds = phmm_counts_lds_bytes(s->max_tmpl, s->max_read, max_bw / 2);
            if (lds > 160 * 1024) return fail(JTK_ERR_UNSUPPORTED, "template + read too long for the LDS staging of phmm_counts_kernel");
            DevPtr d_wide, d_counts, d_lk, d_scratch, d_counter;
            if ((rc = dev_upload(s, d_wide, wide))) return rc;
            if ((rc = dev_alloc<double>(d_counts, (size_t)n_reads * FIT_COUNTS))) return rc;
            if ((rc = dev_alloc<double>(d_lk, n_reads))) return rc;
            if ((rc = dev_alloc<uint32_t>(d_counter, 4))) return rc;
            HIP_TRY(hipMemsetAsync(d_counter.p, 0, 4 * sizeof(uint32_t), s->stream));  // a fresh ticket counter (once per round)
            uint32_t tk_counts = 0;
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, device));
            const uint64_t stride = phmm_counts_scratch_doubles(s->max_tmpl, s->max_read, max_bw / 2);
            uint64_t waves = std::min<uint64_t>(n_reads, (uint64_t)prop.multiProcessorCount * 2);
            waves = std::max<uint64_t>(1, std::min<uint64_t>(waves, (32ull << 30) / (stride * 8)));
            if ((rc = dev_alloc<double>(d_scratch, stride * waves))) return rc;
            launch_phmm_counts(s->stream, s->n_reads, s->d_reads.as<ReadMeta>(), d_wide.as<ChunkMeta>(),
                               s->d_state.as<ChunkState>(), s->bufs, s->d_ey.as<uint8_t>(), s->d_delta.as<uint64_t>(),
                               s->d_hmm2.as<HmmDev>(), d_scratch.as<double>(), stride, (uint32_t)waves,
                               d_counter.as<uint32_t>(), &tk_counts, d_counts.as<double>(), d_lk.as<double>(), s->max_tmpl,
                               s->max_read, max_bw / 2);
            HIP_TRY(hipMemcpyAsync(counts.data(), d_counts.p, counts.size() * 8, hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipMemcpyAsync(lks.data(), d_lk.p, lks.size() * 8, hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            HIP_TRY(hipGetLastError());
        }
        double sum[2][FIT_COUNTS];
        memset(sum, 0, sizeof sum);
        for (size_t g = 0; g < n_reads; g++) {  // reads in order, as the oracle adds them
            const int st = strand[g] ? 0 : 1;
            for (int k = 0; k < FIT_COUNTS; k++) sum[st][k] += counts[g * FIT_COUNTS + k];
        }
        jtk_hmm_t nf, nr;
        fit_mstep(cur.forward, sum[0], nf);
        fit_mstep(cur.reverse, sum[1], nr);
        cur.forward = nf;
        cur.reverse = nr;
        cons.swap(cons2);
        cops.swap(cops2);
        coff.swap(coff2);
        ooff.swap(ooff2);
    }
    *forward_out = cur.forward;
    *reverse_out = cur.reverse;
    return 0;
}

int jtk_lc_modification_table(const jtk_lc_params_t *params, const uint8_t *tmpl, uint64_t tmpl_len,
                              uint32_t n_reads, const uint8_t *read_bases, const uint64_t *read_off,
                              const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, double *table,
                              double *lk, int device) {
    g_last_error.clear();
    if (!table || !lk) return fail(JTK_ERR_INVALID_ARG, "null output");
    jtk_lc_chunk_t ch;
    memset(&ch, 0, sizeof ch);
    ch.chunk_id = 0;
    ch.copy_num = 2;
    ch.n_reads = n_reads;
    ch.tmpl_off = 0;
    ch.tmpl_len = tmpl_len;
    ch.read_first = 0;
    jtk_lc_session_t *s = nullptr;
    int rc = jtk_lc_session_create(params, 1, &ch, tmpl, read_bases, read_off, ops, ops_off, strand, 2, device, &s);
    if (rc) return rc;
    std::unique_ptr<jtk_lc_session> guard(s);
    hipStream_t st = s->stream;
    ChunkState *state = s->d_state.as<ChunkState>();
    launch_reset_pass(st, s->n_chunks, state, s->d_state0.as<ChunkState>(), s->d_nactive.as<uint32_t>(), JTK_NACTIVE_SLOTS);
    launch_band_prep(st, s->n_reads, s->d_reads.as<ReadMeta>(), s->d_chunks.as<ChunkMeta>(), state, s->bufs,
                     s->d_delta.as<uint64_t>(), 0, s->max_tmpl, s->max_read);
    launch_phmm(st, s->n_reads, s->d_reads.as<ReadMeta>(), s->d_chunks.as<ChunkMeta>(), state, s->bufs,
                s->d_ey.as<uint8_t>(), s->d_delta.as<uint64_t>(), s->d_hmm2.as<HmmDev>(), s->stripes->set(), s->n_waves, s->d_counter.as<uint32_t>(), &s->tk_phmm, s->d_raw.as<double>(), s->d_rawG.as<int>(),
                s->d_lk.as<double>(), s->max_tmpl, s->max_read, 0);
    if (s->n_wide_reads)
        launch_phmm_wide(st, s->n_reads, s->d_reads.as<ReadMeta>(), s->d_chunks.as<ChunkMeta>(), state, s->bufs,
                         s->d_ey.as<uint8_t>(), s->d_delta.as<uint64_t>(), s->d_hmm2.as<HmmDev>(),
                         s->d_wide_scratch.as<double>(), s->wide_stride, s->n_wide_waves, s->d_wide_counter.as<uint32_t>(), &s->tk_wide,
                         s->d_raw.as<double>(), s->d_rawG.as<int>(), s->d_lk.as<double>(), s->max_tmpl, s->max_read, 0,
                         s->max_wide_radius);
    launch_finalize(st, s->n_reads, s->d_reads.as<ReadMeta>(), s->d_chunks.as<ChunkMeta>(), state,
                    s->d_hmm2.as<HmmDev>(), s->d_raw.as<double>(), s->d_rawG.as<int>(), s->d_lk.as<double>(), s->max_tmpl, 0);
    ChunkState cs;
    HIP_TRY(hipMemcpyAsync(&cs, state, sizeof cs, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(lk, s->d_lk.p, (size_t)n_reads * 8, hipMemcpyDeviceToHost, st));
    const size_t cols = (size_t)JTK_NUM_ROW * (tmpl_len + 1);
    for (uint32_t r = 0; r < n_reads; r++)
        HIP_TRY(hipMemcpyAsync(table + (size_t)r * cols, s->d_raw.as<double>() + s->h_reads[r].table_off, cols * 8,
                               hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    if (cs.status != 0) return fail(cs.status, "modification table failed (ops mismatch or unsupported band)");
    return 0;
}


// log P(read | template) of every read of a resident batch with a fixed band radius: band_prep + the forward
// sweep of phmm_kernel (the backward sweep runs too and is ignored; these batches are tiny).  Used by the
// gains calibration (gains.hip), which evaluates kiley's likelihood_antidiagonal_bootstrap 180,000 times.
int jtk_internal_likelihoods(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_chunk_t *chunks,
                             const uint8_t *tmpl_bases, const uint8_t *read_bases, const uint64_t *read_off,
                             const uint8_t *ops, const uint64_t *ops_off, const uint8_t *strand, uint32_t radius,
                             int device, double *lk_out) {
    std::vector<ChunkExtra> extra(n_chunks);
    for (auto &e : extra) {
        memset(&e, 0, sizeof e);
        e.radius = radius;
    }
    jtk_lc_session_t *s = nullptr;
    uint32_t stride = 1;
    for (size_t c = 0; c < n_chunks; c++) stride = std::max(stride, std::min<uint32_t>(chunks[c].copy_num, JTK_MAX_COPY));
    int rc = session_create_ex(params, n_chunks, chunks, tmpl_bases, read_bases, read_off, ops, ops_off, strand, stride, device,
                               extra.data(), 0, &s);
    if (rc) return rc;
    std::unique_ptr<jtk_lc_session> guard(s);
    hipStream_t st = s->stream;
    ChunkState *state = s->d_state.as<ChunkState>();
    launch_reset_pass(st, s->n_chunks, state, s->d_state0.as<ChunkState>(), s->d_nactive.as<uint32_t>(), JTK_NACTIVE_SLOTS);
    launch_band_prep(st, s->n_reads, s->d_reads.as<ReadMeta>(), s->d_chunks.as<ChunkMeta>(), state, s->bufs,
                     s->d_delta.as<uint64_t>(), 0, s->max_tmpl, s->max_read);
    launch_phmm(st, s->n_reads, s->d_reads.as<ReadMeta>(), s->d_chunks.as<ChunkMeta>(), state, s->bufs,
                s->d_ey.as<uint8_t>(), s->d_delta.as<uint64_t>(), s->d_hmm2.as<HmmDev>(), s->stripes->set(), s->n_waves, s->d_counter.as<uint32_t>(), &s->tk_phmm, s->d_raw.as<double>(), s->d_rawG.as<int>(),
                s->d_lk.as<double>(), s->max_tmpl, s->max_read, 0);
    if (s->n_wide_reads)
        launch_phmm_wide(st, s->n_reads, s->d_reads.as<ReadMeta>(), s->d_chunks.as<ChunkMeta>(), state, s->bufs,
                         s->d_ey.as<uint8_t>(), s->d_delta.as<uint64_t>(), s->d_hmm2.as<HmmDev>(),
                         s->d_wide_scratch.as<double>(), s->wide_stride, s->n_wide_waves, s->d_wide_counter.as<uint32_t>(), &s->tk_wide,
                         s->d_raw.as<double>(), s->d_rawG.as<int>(), s->d_lk.as<double>(), s->max_tmpl, s->max_read, 0,
                         s->max_wide_radius);
    std::vector<ChunkState> cs(n_chunks);
    HIP_TRY(hipMemcpyAsync(cs.data(), state, cs.size() * sizeof(ChunkState), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(lk_out, s->d_lk.p, (size_t)s->n_reads * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    for (const ChunkState &c : cs)
        if (c.status != 0) return fail(c.status, "likelihood batch failed (ops mismatch or unsupported band)");
    return 0;
}

int jtk_lc_cluster_features(const jtk_lc_params_t *params, size_t n_chunks, const jtk_lc_feature_chunk_t *chunks,
                            const double *variants, const uint32_t *variant_type, uint32_t *label, double *log_post,
                            uint32_t post_stride, jtk_lc_result_t *result, int device) {
    g_last_error.clear();
    if (!params || (n_chunks && (!chunks || !variants || !variant_type || !label || !log_post || !result)))
        return fail(JTK_ERR_INVALID_ARG, "null argument");
    int rc = pick_device(device);
    if (rc) return rc;
    // declared before the session: destructors run in reverse order, so the session's (which synchronises its stream)
    // runs before these blocks go back to the pool on every early return
    DevPtr d_params, d_chunks, d_state, d_var, d_vt, d_vtoff, d_label, d_post, d_lg, d_lgoff;
    jtk_lc_session sess;
    jtk_lc_session *s = &sess;
    s->device = device;
    HIP_TRY(hipStreamCreate(&s->stream));
    std::vector<ChunkMeta> cms(n_chunks);
    std::vector<ChunkState> sts(n_chunks);
    std::vector<uint64_t> vt_off(n_chunks), lg_off(n_chunks);
    uint64_t n_reads = 0, n_var = 0, n_vt = 0, lgo = 0;
    uint32_t max_n = 1, max_d = 1, max_k = 2;
    for (size_t c = 0; c < n_chunks; c++) {
        const jtk_lc_feature_chunk_t &fc = chunks[c];
        if (fc.copy_num > max_k && fc.copy_num <= JTK_MAX_COPY) max_k = fc.copy_num;
        if (fc.read_first != n_reads) return fail(JTK_ERR_INVALID_ARG, "chunks must list their reads contiguously in order");
        memset(&cms[c], 0, sizeof(ChunkMeta));
        memset(&sts[c], 0, sizeof(ChunkState));
        cms[c].chunk_id = fc.chunk_id;
        cms[c].copy_num = fc.copy_num;
        cms[c].n_reads = fc.n_reads;
        cms[c].read_first = (uint32_t)n_reads;
        cms[c].feat_off = fc.var_off;
        cms[c].local_coverage = fc.local_coverage;
        sts[c].dim = fc.dim;
        sts[c].k = 1;
        if (fc.dim > JTK_MAX_DIM || fc.copy_num > JTK_MAX_COPY) sts[c].status = JTK_ERR_UNSUPPORTED;
        else if (fc.copy_num > post_stride)  // a posterior row holds up to copy_num entries
            return fail(JTK_ERR_INVALID_ARG, "post_stride smaller than a chunk's copy_num");
        vt_off[c] = fc.vt_off;
        lg_off[c] = lgo;
        lgo += (uint64_t)fc.n_reads * (JTK_MAX_COPY + 1);
        n_reads += fc.n_reads;
        if (fc.var_off + (uint64_t)fc.n_reads * fc.dim > n_var) n_var = fc.var_off + (uint64_t)fc.n_reads * fc.dim;
        if (fc.vt_off + fc.dim > n_vt) n_vt = fc.vt_off + fc.dim;
    }
    // chunks whose work area fits a CU's LDS run in the table-driven kernels (one launch sized for their maxima); the others --
    // more than JTK_MAX_PILEUP reads, or too large a feature matrix -- in mcmc_kernel_huge with a global-memory work area
    std::vector<uint32_t> in_lds, in_ws;
    for (size_t c = 0; c < n_chunks; c++) {
        if (sts[c].status != 0) continue;
        const jtk_lc_feature_chunk_t &fc = chunks[c];
        const uint32_t d = std::max<uint32_t>(1, fc.dim), k = std::max<uint32_t>(2, fc.copy_num);
        (fc.n_reads > JTK_MAX_PILEUP || mcmc_lds_bytes(std::max<uint32_t>(1, fc.n_reads), d, k) > 160 * 1024 ? in_ws : in_lds).push_back((uint32_t)c);
    }
    for (;;) {
        max_n = max_d = 1;
        for (uint32_t c : in_lds) {
            max_n = std::max(max_n, chunks[c].n_reads);
            max_d = std::max(max_d, chunks[c].dim);
        }
        if (in_lds.empty() || mcmc_lds_bytes(max_n, max_d, max_k) <= 160 * 1024) break;
        auto worst = std::max_element(in_lds.begin(), in_lds.end(), [&](uint32_t a, uint32_t b) {
            return (uint64_t)chunks[a].n_reads * std::max<uint32_t>(1, chunks[a].dim) < (uint64_t)chunks[b].n_reads * std::max<uint32_t>(1, chunks[b].dim);
        });
        in_ws.push_back(*worst);  // the maxima of the launch combine beyond a CU's LDS: its largest member leaves
        in_lds.erase(worst);
    }
    std::sort(in_ws.begin(), in_ws.end());
    std::vector<jtk_lc_params_t> pv(1, *params);
    std::vector<double> varv(variants, variants + n_var);
    std::vector<uint32_t> vtv(variant_type, variant_type + 2 * n_vt);
    if ((rc = dev_upload(s, d_params, pv))) return rc;
    if ((rc = dev_upload(s, d_chunks, cms))) return rc;
    if ((rc = dev_upload(s, d_state, sts))) return rc;
    if ((rc = dev_upload(s, d_var, varv))) return rc;
    if ((rc = dev_upload(s, d_vt, vtv))) return rc;
    if ((rc = dev_upload(s, d_vtoff, vt_off))) return rc;
    if ((rc = dev_upload(s, d_lgoff, lg_off))) return rc;
    if ((rc = dev_alloc<uint32_t>(d_label, n_reads))) return rc;
    if ((rc = dev_alloc<double>(d_post, n_reads * post_stride))) return rc;
    if ((rc = dev_alloc<double>(d_lg, lgo))) return rc;
    DevPtr d_split, d_order, d_ws, d_wsoff;  // light / general chunk lists of the chain launch; the two launches' chunk lists
    std::vector<uint32_t> order(in_lds);
    order.insert(order.end(), in_ws.begin(), in_ws.end());
    for (size_t c = 0; c < n_chunks; c++)  // (chunks that failed validation: listed too, they return at once)
        if (sts[c].status != 0) order.insert(order.begin() + (ptrdiff_t)in_lds.size(), (uint32_t)c);
    const uint32_t n_lds = (uint32_t)(order.size() - in_ws.size());
    if ((rc = dev_upload(s, d_order, order))) return rc;
    std::vector<uint64_t> ws_off;
    uint64_t ws = 0;
    uint32_t hn = 1, hd = 1;
    for (uint32_t c : in_ws) {
        ws_off.push_back(ws);
        ws += mcmc_ws_bytes(std::max<uint32_t>(1, chunks[c].n_reads), std::max<uint32_t>(1, chunks[c].dim), std::max<uint32_t>(2, chunks[c].copy_num));
        hn = std::max(hn, chunks[c].n_reads);
        hd = std::max(hd, chunks[c].dim);
    }
    if (!in_ws.empty()) {
        if ((rc = dev_alloc<uint8_t>(d_ws, ws))) return rc;
        if ((rc = dev_upload(s, d_wsoff, ws_off))) return rc;
    }
    hipEvent_t ev0, ev1;
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, s->stream));
    if ((rc = dev_alloc<uint32_t>(d_split, 2 * n_chunks + 8))) return rc;
    if (n_lds && launch_mcmc(s->stream, n_lds, d_chunks.as<ChunkMeta>(), d_state.as<ChunkState>(),
                    d_params.as<jtk_lc_params_t>(), d_var.as<double>(), d_vt.as<uint32_t>(), d_vtoff.as<uint64_t>(), 1,
                    d_label.as<uint32_t>(), d_post.as<double>(), post_stride, d_lg.as<double>(), d_lgoff.as<uint64_t>(),
                    max_n, max_d, max_k, nullptr, d_order.as<uint32_t>(), d_split.as<uint32_t>(), nullptr, nullptr, nullptr) != 0)
        return fail(JTK_ERR_INTERNAL, "the chain kernel could not be launched (jump table upload failed)");
    if (!in_ws.empty() &&
        launch_mcmc_huge(s->stream, (uint32_t)in_ws.size(), d_chunks.as<ChunkMeta>(), d_state.as<ChunkState>(),
                         d_params.as<jtk_lc_params_t>(), d_var.as<double>(), d_vt.as<uint32_t>(), d_vtoff.as<uint64_t>(), 1,
                         d_label.as<uint32_t>(), d_post.as<double>(), post_stride, d_lg.as<double>(), d_lgoff.as<uint64_t>(), hn, hd,
                         max_k, nullptr, d_order.as<uint32_t>() + n_lds, d_ws.as<uint8_t>(), d_wsoff.as<uint64_t>()) != 0)
        return fail(JTK_ERR_INTERNAL, "the chain kernel could not be launched (jump table upload failed)");
    HIP_TRY(hipEventRecord(ev1, s->stream));
    HIP_TRY(hipMemcpyAsync(sts.data(), d_state.p, sts.size() * sizeof(ChunkState), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(label, d_label.p, n_reads * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(log_post, d_post.p, n_reads * post_stride * 8, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    memset(&g_timing, 0, sizeof g_timing);
    g_timing.total_ms = ms;
    g_timing.kernel_ms[JTK_K_MCMC] = ms;
    g_timing.kernel_launches[JTK_K_MCMC] = 1;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    int any_fail = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        result[c].score = sts[c].status == 0 ? sts[c].score : 0.0;
        result[c].cluster_num = sts[c].status == 0 ? sts[c].k : 1;
        result[c].status = sts[c].status;
        result[c].polish_rounds = 0;
        result[c].n_variants = sts[c].dim;
        if (sts[c].status != 0) any_fail = 1;
    }
    return any_fail ? fail(JTK_ERR_CHUNK_FAILED, "at least one chunk failed; see result[].status") : 0;
}

int jtk_lc_trim_cache(int device) {
    if (device < 0 || device >= JTK_POOL_DEVICES) return JTK_ERR_INVALID_ARG;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return JTK_ERR_NO_DEVICE;
    (void)hipSetDevice(device);
    {
        std::lock_guard<std::mutex> lock(g_stripe_mutex);
        g_stripes[device].reset();  // running sessions keep the set they hold
    }
    g_pool.trim(device);
    (void)hipSetDevice(cur);
    return 0;
}

const char *jtk_lc_last_error(void) { return g_last_error.c_str(); }
void jtk_internal_set_error(const char *msg) { g_last_error = msg ? msg : ""; }

int jtk_lc_last_timing(jtk_lc_timing_t *out) {
    if (!out) return JTK_ERR_INVALID_ARG;
    *out = g_timing;
    return 0;
}

int jtk_lc_device_ok(int device) {
    const std::string keep = g_last_error;
    const int rc = pick_device(device);
    g_last_error = keep;
    return rc == 0 ? 1 : 0;
}

}  // extern "C"
