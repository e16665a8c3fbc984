// local_clustering.hpp -- C++ host-side mirror of the reference's stage surface, on top of the C ABI.
//
//   pub trait LocalClustering { fn local_clustering(&mut self);
//                               fn local_clustering_selected(&mut self, selection: &HashSet<u64>); }
//   impl LocalClustering for DataSet                          haplotyper/src/local_clustering/mod.rs:17-30
//
// Same names, same argument meaning, same effects on the DataSet (Node.{cluster,posterior,cigar},
// Chunk.{seq,score,cluster_num}, Coverage), same error behaviour (the reference panics; this throws).  The types
// are the subset of definitions/src/lib.rs the stage touches.  Rust is not available in the build image, so
// this header is how the host logic around the ABI (mod.rs:33-83, 244-260; normalize.rs; misc.rs:177-225,
// 394-407) is exercised end to end (tests/test_host_mirror.py).
//
// The preambles of local_clustering_selected:
//   ds.update_models_on_both_strands()   (mod.rs:58, model_tune.rs:96-156)  update_models_on_both_strands below: training
//                                                                            pile-ups picked here, refit by jtk_lc_fit_model
//                                                                            (LocalClusteringOptions::refit_model = false
//                                                                            keeps ds.model_param as given)
//   estimate_gain_default(&hmm)          (mod.rs:60)                        jtk_lc_estimate_gains on the device, unless
//                                                                            LocalClusteringOptions::gains supplies them
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <ostream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "jtk_lc.h"

namespace jtk {

enum class ReadType { CCS, CLR, ONT, None };  // definitions/src/lib.rs:156-162

struct Op {  // definitions/src/lib.rs:816-823
    enum Kind { Match, Del, Ins } kind;
    size_t len;
};
using Ops = std::vector<Op>;

struct Node {  // definitions/src/lib.rs:672-683
    size_t position_from_start = 0;
    uint64_t chunk = 0;
    uint64_t cluster = 0;
    std::string seq;  // upper-case ACGT, already in chunk orientation
    bool is_forward = true;
    Ops cigar;
    std::vector<double> posterior;
};

struct EncodedRead {  // definitions/src/lib.rs (nodes only)
    uint64_t id = 0;
    std::vector<Node> nodes;
};

struct Chunk {  // definitions/src/lib.rs:403-415
    uint64_t id = 0;
    std::string seq;
    size_t cluster_num = 1;
    size_t copy_num = 2;
    double score = 0.0;
};

struct Coverage {  // definitions/src/lib.rs:46-93
    enum Kind { NotAvailable, Protected, Estimated } kind = NotAvailable;
    double value = 0.0;
    bool is_protected() const { return kind == Protected; }
    void set(double v) {
        kind = Estimated;
        value = v;
    }
    double unwrap() const {
        if (kind == NotAvailable) throw std::runtime_error("Coverage::unwrap on NotAvailable");
        return value;
    }
};

struct HMMParamOnStrands {  // definitions/src/lib.rs:95-99
    jtk_hmm_t forward, reverse;
};

struct DataSet {  // definitions/src/lib.rs:6-34 (fields the stage reads or writes)
    Coverage coverage;
    ReadType read_type = ReadType::ONT;
    std::vector<Chunk> selected_chunks;
    std::vector<EncodedRead> encoded_reads;
    HMMParamOnStrands model_param;
};

struct LocalClusteringOptions {
    int device = 0;
    const jtk_gains_t *gains = nullptr;  // optional: a cached result of estimate_gain_default (likelihood_gains.rs:186-192)
    // The reference panics when a chunk hits one of its asserts.  nullptr: this mirror throws like it, before touching the
    // DataSet.  Otherwise chunks that come back with a status (where the reference would panic, or a shape this build does
    // not take) are left exactly as they were, their (chunk id, jtk_status) pairs are appended here and every other chunk
    // is written back.
    std::vector<std::pair<uint64_t, int>> *failed = nullptr;
    bool refit_model = true;  // run update_models_on_both_strands (mod.rs:58) before clustering, as the reference does
    // The reference's per-chunk `debug!("RECORD\t{chunk_id}\t{elapsed}\t{polished_time}\t{len}\t{score:.3}\t{cov}")`
    // (mod.rs:121), one line per chunk that was written back.  The device runs the chunks of a call together, so a chunk's
    // milliseconds are its share of the call's kernel time (jtk_lc_last_timing): pair-HMM + polishing by band cells x passes,
    // the chain by proposals x candidate k.  nullptr: no lines.
    std::ostream *record = nullptr;
    // The reference's trace! rows of every clustered chunk (log level Trace: TOTAL / CAND / PICK / DUMP / RANGE / LK / COUNTS,
    // pseudo_mcmc.rs:122-127,236,250-262,467-472,539), chunk after chunk in chunk-id order: the call's chunks once more through a
    // resident session and jtk_lc_session_trace (a debugging aid, like the log level it mirrors).  Not available for a call that
    // holds a chunk of copy number >= 8 (clustering_recursive): nothing is written then.  nullptr: no rows.
    std::ostream *trace = nullptr;
};

inline double band_frac(ReadType t) {  // definitions/src/lib.rs:173-175, 201-210
    switch (t) {
        case ReadType::CCS: return 0.01;
        case ReadType::ONT: return 0.03;
        default: return 0.05;
    }
}

// misc.rs:177-186
inline std::vector<uint8_t> ops_to_kiley(const Ops &ops) {
    std::vector<uint8_t> out;
    for (const Op &op : ops)
        out.insert(out.end(), op.len,
                   (uint8_t)(op.kind == Op::Match ? JTK_OP_MATCH : (op.kind == Op::Del ? JTK_OP_DEL : JTK_OP_INS)));
    return out;
}

// misc.rs:188-225: Match and Mismatch merge into one M run
inline Ops kiley_op_to_ops(const uint8_t *k, size_t n) {
    Ops ops;
    for (size_t i = 0; i < n; i++) {
        const Op::Kind kd = k[i] == JTK_OP_DEL ? Op::Del : (k[i] == JTK_OP_INS ? Op::Ins : Op::Match);
        if (!ops.empty() && ops.back().kind == kd)
            ops.back().len++;
        else
            ops.push_back(Op{kd, 1});
    }
    return ops;
}

// misc.rs:394-407
inline void update_coverage(DataSet &ds) {
    if (ds.coverage.is_protected()) return;
    std::unordered_map<uint64_t, uint32_t> counts;
    for (const auto &r : ds.encoded_reads)
        for (const auto &n : r.nodes) counts[n.chunk]++;
    if (counts.empty()) throw std::runtime_error("update_coverage: no nodes");
    std::vector<uint32_t> c;
    for (const auto &kv : counts) c.push_back(kv.second);
    std::sort(c.begin(), c.end());
    ds.coverage.set((double)c[c.size() / 2] / 2.0);
}

// normalize.rs:6-51 -- walks every node of the dataset, as the reference does
inline void normalize_local_clustering(DataSet &ds) {
    std::unordered_map<uint64_t, std::vector<Node *>> pileups;
    for (auto &r : ds.encoded_reads)
        for (auto &n : r.nodes) pileups[n.chunk].push_back(&n);
    std::unordered_map<uint64_t, size_t> cluster_num;
    for (const auto &c : ds.selected_chunks) cluster_num[c.id] = c.cluster_num;
    for (auto &kv : pileups) {
        auto it = cluster_num.find(kv.first);
        if (it == cluster_num.end()) throw std::runtime_error("normalize_local_clustering: node on an unknown chunk");
        const size_t k = it->second;
        auto &nodes = kv.second;
        std::vector<uint32_t> label(nodes.size());
        std::vector<double> post(nodes.size() * (k ? k : 1));
        for (size_t i = 0; i < nodes.size(); i++) {
            if (nodes[i]->posterior.size() != k)  // assert_eq!(n.posterior.len(), max_cluster) normalize.rs:27-29
                throw std::runtime_error("normalize_local_clustering: posterior length != cluster_num");
            label[i] = (uint32_t)nodes[i]->cluster;
            std::copy(nodes[i]->posterior.begin(), nodes[i]->posterior.end(), post.begin() + i * k);
        }
        if (k == 0) continue;
        const int rc = jtk_lc_normalize_pileup((uint32_t)nodes.size(), (uint32_t)k, label.data(), post.data(), (uint32_t)k);
        if (rc != 0) throw std::runtime_error(std::string("normalize_local_clustering: ") + jtk_lc_strerror(rc));
        for (size_t i = 0; i < nodes.size(); i++) {
            nodes[i]->cluster = label[i];
            nodes[i]->posterior.assign(post.begin() + i * k, post.begin() + (i + 1) * k);
        }
    }
}

// ModelFit::update_models_on_both_strands (model_tune.rs:20-25 -> estimate_model_parameters_on_both_strands :96-156).
// Training pile-ups (:99-118): every selected chunk's nodes in order of appearance; keep the pile-ups whose size is within
// [max(cov, 2) - 2, cov + 2) of cov = the (len / 2)-th smallest size, sort them by chunk id, take the first
// TRAIN_UNIT_SIZE = 5.  TRAIN_ROUND = 10 rounds of polish + fit run inside jtk_lc_fit_model.
inline void update_models_on_both_strands(DataSet &ds, int device = 0) {
    const size_t TRAIN_UNIT_SIZE = 5;
    const uint32_t TRAIN_ROUND = 10;
    std::unordered_map<uint64_t, std::vector<const Node *>> pileups;
    std::unordered_map<uint64_t, const Chunk *> chunk_of;
    for (const auto &c : ds.selected_chunks) {
        pileups[c.id];
        chunk_of[c.id] = &c;
    }
    for (const auto &r : ds.encoded_reads)
        for (const auto &n : r.nodes) pileups[n.chunk].push_back(&n);
    std::vector<size_t> covs;
    for (const auto &kv : pileups) covs.push_back(kv.second.size());
    if (covs.empty()) throw std::runtime_error("update_models_on_both_strands: no pile-up");
    std::nth_element(covs.begin(), covs.begin() + covs.size() / 2, covs.end());
    const size_t cov = covs[covs.size() / 2];
    std::vector<uint64_t> ids;
    for (const auto &kv : pileups) {
        const size_t len = kv.second.size();
        if (len >= std::max<size_t>(cov, 2) - 2 && len < cov + 2) ids.push_back(kv.first);
    }
    // model_tune.rs:99-118: sort by id, truncate to TRAIN_UNIT_SIZE, and only THEN drop the pile-ups of chunks that are not
    // among the selected chunks (the filter_map of :119-122) -- a stray node can cost a training pile-up
    std::sort(ids.begin(), ids.end());
    if (ids.size() > TRAIN_UNIT_SIZE) ids.resize(TRAIN_UNIT_SIZE);
    ids.erase(std::remove_if(ids.begin(), ids.end(), [&](uint64_t id) { return chunk_of.count(id) == 0; }), ids.end());
    std::vector<jtk_lc_chunk_t> chunks;
    std::vector<uint8_t> tmpl, reads, ops, strand;
    std::vector<uint64_t> read_off{0}, ops_off{0};
    for (uint64_t id : ids) {
        const Chunk *c = chunk_of[id];
        jtk_lc_chunk_t ch;
        ch.chunk_id = id;
        ch.copy_num = (uint32_t)c->copy_num;
        ch.n_reads = (uint32_t)pileups[id].size();
        ch.tmpl_off = tmpl.size();
        ch.tmpl_len = c->seq.size();
        ch.read_first = strand.size();
        chunks.push_back(ch);
        tmpl.insert(tmpl.end(), c->seq.begin(), c->seq.end());
        for (const Node *n : pileups[id]) {
            reads.insert(reads.end(), n->seq.begin(), n->seq.end());
            read_off.push_back(reads.size());
            const std::vector<uint8_t> k = ops_to_kiley(n->cigar);
            ops.insert(ops.end(), k.begin(), k.end());
            ops_off.push_back(ops.size());
            strand.push_back(n->is_forward ? 1 : 0);
        }
    }
    if (chunks.empty() || strand.empty()) throw std::runtime_error("update_models_on_both_strands: no training pile-up");  // :135
    jtk_lc_params_t params{};
    params.forward = ds.model_param.forward;
    params.reverse = ds.model_param.reverse;
    params.band_frac = band_frac(ds.read_type);
    jtk_hmm_t f, r;
    const int rc = jtk_lc_fit_model(&params, chunks.size(), chunks.data(), tmpl.data(), reads.data(), read_off.data(), ops.data(),
                                    ops_off.data(), strand.data(), TRAIN_ROUND, &f, &r, device);
    if (rc != 0) throw std::runtime_error(std::string("update_models_on_both_strands: ") + jtk_lc_strerror(rc) + ": " + jtk_lc_last_error());
    ds.model_param.forward = f;
    ds.model_param.reverse = r;
}

// mod.rs:56-83
inline void local_clustering_selected(DataSet &ds, const std::unordered_set<uint64_t> &selection,
                                      const LocalClusteringOptions &opt = LocalClusteringOptions()) {
    update_coverage(ds);  // mod.rs:57
    if (opt.refit_model) update_models_on_both_strands(ds, opt.device);  // mod.rs:58
    jtk_lc_params_t params;
    params.forward = ds.model_param.forward;  // mod.rs:59 (refit of mod.rs:58 is the caller's, see header)
    params.reverse = ds.model_param.reverse;
    if (opt.gains) {
        params.gains = *opt.gains;
    } else {  // mod.rs:60: estimate_gain_default = estimate_gain(hmm, 309423, 100, 10, 3)
        const int rc = jtk_lc_estimate_gains(&params.forward, &params.reverse, 309423, 100, 10, 3, &params.gains, opt.device);
        if (rc != 0) throw std::runtime_error(std::string("jtk_lc_estimate_gains: ") + jtk_lc_last_error());
    }
    params.haploid_coverage = ds.coverage.unwrap();
    params.band_frac = band_frac(ds.read_type);
    // pileup_nodes (mod.rs:33-53): nodes of the selected chunks in order of appearance, then a STABLE sort by
    // the number of non-'|' alignment columns against the unpolished chunk sequence
    struct Pile {
        const Chunk *chunk;
        std::vector<Node *> nodes;
    };
    std::unordered_map<uint64_t, Pile> piles;
    for (const auto &c : ds.selected_chunks)
        if (selection.count(c.id)) piles[c.id] = Pile{&c, {}};
    for (auto &r : ds.encoded_reads)
        for (auto &n : r.nodes) {
            auto it = piles.find(n.chunk);
            if (it != piles.end()) it->second.nodes.push_back(&n);
        }
    std::vector<uint64_t> order;
    for (auto &kv : piles)
        if (!kv.second.nodes.empty()) order.push_back(kv.first);  // .filter(|(_, (nodes, _))| !nodes.is_empty())
    std::sort(order.begin(), order.end());
    std::vector<jtk_lc_chunk_t> chunks;
    std::vector<uint8_t> tmpl, reads, ops, strand;
    std::vector<uint64_t> read_off{0}, ops_off{0};
    for (uint64_t id : order) {
        Pile &p = piles[id];
        std::vector<std::vector<uint8_t>> kops(p.nodes.size());
        std::vector<std::pair<uint64_t, size_t>> keyed(p.nodes.size());
        for (size_t i = 0; i < p.nodes.size(); i++) {
            kops[i] = ops_to_kiley(p.nodes[i]->cigar);
            uint64_t key = 0;
            const int rc = jtk_lc_pileup_sort_key((const uint8_t *)p.chunk->seq.data(), p.chunk->seq.size(),
                                                  (const uint8_t *)p.nodes[i]->seq.data(), p.nodes[i]->seq.size(),
                                                  kops[i].data(), kops[i].size(), &key);
            if (rc != 0) throw std::runtime_error(std::string("pileup_nodes: ") + jtk_lc_strerror(rc));
            keyed[i] = {key, i};
        }
        std::stable_sort(keyed.begin(), keyed.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        std::vector<Node *> sorted_nodes;
        jtk_lc_chunk_t ch;
        ch.chunk_id = p.chunk->id;
        ch.copy_num = (uint32_t)p.chunk->copy_num;
        ch.n_reads = (uint32_t)p.nodes.size();
        ch.tmpl_off = tmpl.size();
        ch.tmpl_len = p.chunk->seq.size();
        ch.read_first = strand.size();
        chunks.push_back(ch);
        tmpl.insert(tmpl.end(), p.chunk->seq.begin(), p.chunk->seq.end());
        for (const auto &ki : keyed) {
            Node *n = p.nodes[ki.second];
            sorted_nodes.push_back(n);
            reads.insert(reads.end(), n->seq.begin(), n->seq.end());
            read_off.push_back(reads.size());
            ops.insert(ops.end(), kops[ki.second].begin(), kops[ki.second].end());
            ops_off.push_back(ops.size());
            strand.push_back(n->is_forward ? 1 : 0);
        }
        p.nodes.swap(sorted_nodes);
    }
    // the hot loop of mod.rs:64-72, on the GPU
    uint32_t stride = 1;
    for (const auto &c : chunks) stride = std::max(stride, c.copy_num);
    const size_t n = strand.size();
    std::vector<uint32_t> label(n);
    std::vector<double> post(n * stride);
    std::vector<jtk_lc_result_t> result(chunks.size());
    const uint64_t cons_cap = 2 * tmpl.size() + 64 * chunks.size() + 64, ops_cap = 2 * ops.size() + 64 * n + 64;
    std::vector<uint8_t> cons(cons_cap), ops_out(ops_cap);
    std::vector<uint64_t> cons_off(chunks.size() + 1), ops_out_off(n + 1);
    const int rc = jtk_lc_cluster_chunks(&params, chunks.size(), chunks.data(), tmpl.data(), reads.data(), read_off.data(),
                                         ops.data(), ops_off.data(), strand.data(), label.data(), post.data(), stride,
                                         result.data(), cons.data(), cons_off.data(), cons_cap, ops_out.data(),
                                         ops_out_off.data(), ops_cap, opt.device);
    if (rc != 0 && !(rc == JTK_ERR_CHUNK_FAILED && opt.failed))  // the reference panics on every failure of this stage
        throw std::runtime_error(std::string("local_clustering: ") + jtk_lc_strerror(rc) + ": " + jtk_lc_last_error());
    // update_by_clusterings (mod.rs:244-260) and the chunk write-back (mod.rs:74-81)
    std::unordered_map<uint64_t, size_t> index_of;
    for (size_t c = 0; c < order.size(); c++) index_of[order[c]] = c;
    for (size_t c = 0; c < order.size(); c++) {
        if (result[c].status != 0) {
            opt.failed->emplace_back(order[c], result[c].status);
            continue;
        }
        Pile &p = piles[order[c]];
        const size_t k = result[c].cluster_num;
        for (size_t r = 0; r < p.nodes.size(); r++) {
            const size_t g = chunks[c].read_first + r;
            Node *node = p.nodes[r];
            node->posterior.assign(post.begin() + g * stride, post.begin() + g * stride + k);
            node->cluster = label[g];
            node->cigar = kiley_op_to_ops(ops_out.data() + ops_out_off[g], ops_out_off[g + 1] - ops_out_off[g]);
        }
    }
    for (auto &chunk : ds.selected_chunks) {
        auto it = index_of.find(chunk.id);
        if (it == index_of.end()) continue;
        const size_t c = it->second;
        if (result[c].status != 0) continue;
        chunk.seq.assign((const char *)cons.data() + cons_off[c], cons_off[c + 1] - cons_off[c]);
        chunk.score = result[c].score;
        chunk.cluster_num = result[c].cluster_num;
    }
    if (opt.record) {  // mod.rs:121
        jtk_lc_timing_t tm;
        if (jtk_lc_last_timing(&tm) == 0) {
            std::vector<double> w_dp(chunks.size()), w_mc(chunks.size());
            double s_dp = 0.0, s_mc = 0.0;
            for (size_t c = 0; c < chunks.size(); c++) {
                const double nr = chunks[c].n_reads;
                const double bases = (double)(read_off[chunks[c].read_first + chunks[c].n_reads] - read_off[chunks[c].read_first]);
                const double passes = std::min<double>(result[c].polish_rounds + 1.0, 21.0);
                w_dp[c] = passes * (nr * chunks[c].tmpl_len + bases);
                const int64_t kt = std::max<int64_t>(1, std::min<int64_t>(chunks[c].copy_num, 1 + 2 * (int64_t)result[c].n_variants) - 1);
                w_mc[c] = result[c].n_variants > 0 ? nr * (double)kt : 0.0;
                s_dp += w_dp[c];
                s_mc += w_mc[c];
            }
            for (size_t c = 0; c < chunks.size(); c++) {
                if (result[c].status != 0) continue;
                const double polish_ms = (tm.kernel_ms[JTK_K_PHMM] + tm.kernel_ms[JTK_K_POLISH]) * w_dp[c] / std::max(s_dp, 1.0);
                const double elapsed = polish_ms + tm.kernel_ms[JTK_K_FILTER] / (double)chunks.size() +
                                       tm.kernel_ms[JTK_K_MCMC] * w_mc[c] / std::max(s_mc, 1.0);
                char line[256];
                snprintf(line, sizeof line, "RECORD\t%llu\t%.3f\t%.3f\t%llu\t%.3f\t%u\n", (unsigned long long)order[c], elapsed,
                         polish_ms, (unsigned long long)(cons_off[c + 1] - cons_off[c]), result[c].score, chunks[c].n_reads);
                *opt.record << line;
            }
        }
    }
    if (opt.trace) {
        jtk_lc_session_t *sess = nullptr;
        if (jtk_lc_session_create(&params, chunks.size(), chunks.data(), tmpl.data(), reads.data(), read_off.data(), ops.data(),
                                  ops_off.data(), strand.data(), stride, opt.device, &sess) == 0) {
            const int run_rc = jtk_lc_session_run(sess, 0);
            if (run_rc == 0 || run_rc == JTK_ERR_CHUNK_FAILED) {
                std::vector<char> text(1 << 16);
                for (size_t c = 0; c < chunks.size(); c++) {
                    if (result[c].status != 0) continue;
                    size_t len = 0;
                    int trc = jtk_lc_session_trace(sess, c, text.data(), text.size(), &len);
                    if (trc != 0 && len > text.size()) {
                        text.resize(len);
                        trc = jtk_lc_session_trace(sess, c, text.data(), text.size(), &len);
                    }
                    if (trc != 0) break;  // (JTK_ERR_UNSUPPORTED: the call holds a chunk of copy number >= 8)
                    opt.trace->write(text.data(), (std::streamsize)len);
                }
            }
            jtk_lc_session_destroy(sess);
        }
    }
    normalize_local_clustering(ds);  // mod.rs:82
}

// mod.rs:23-26
inline void local_clustering(DataSet &ds, const LocalClusteringOptions &opt = LocalClusteringOptions()) {
    std::unordered_set<uint64_t> selection;
    for (const auto &c : ds.selected_chunks) selection.insert(c.id);
    local_clustering_selected(ds, selection, opt);
}

}  // namespace jtk
