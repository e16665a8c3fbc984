// Test driver of the C++ host mirror: builds a DataSet from synthetic pile-ups (generation order, nodes spread
// over reads), runs jtk::local_clustering on the GPU and dumps what the stage wrote back.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "host/local_clustering.hpp"
#include "jtk_synth.h"

static std::string cigar_string(const jtk::Ops &ops) {  // Display for Ops, definitions/src/lib.rs:825-838
    std::string s;
    for (const auto &op : ops) {
        s += std::to_string(op.len);
        s += op.kind == jtk::Op::Match ? 'M' : (op.kind == jtk::Op::Del ? 'D' : 'I');
    }
    return s;
}

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s n_chunks tmpl_len reads_per_hap gains_file(18 doubles; '-' = estimate on the device) n_selected [refit]\n", argv[0]);
        return 2;
    }
    const int n_chunks = atoi(argv[1]), tmpl_len = atoi(argv[2]), rph = atoi(argv[3]), n_selected = atoi(argv[5]);
    jtk_gains_t gains{};
    gains.max_homopolymer_len = 3;
    const bool device_gains = std::string(argv[4]) == "-";  // mod.rs:60 inside the stage call
    if (!device_gains) {
        FILE *f = fopen(argv[4], "r");
        if (!f) return 3;
        jtk_gain_profile_t *rows[3] = {gains.subst, gains.deletions, gains.insertions};
        for (int t = 0; t < 3; t++)
            for (int h = 0; h < 3; h++)
                if (fscanf(f, "%lf %lf", &rows[t][h].gain, &rows[t][h].prob) != 2) return 3;
        fclose(f);
    }
    jtk::DataSet ds;
    ds.read_type = jtk::ReadType::ONT;
    ds.coverage.kind = jtk::Coverage::Protected;  // like `-c` on the CLI: keep the caller's haploid coverage
    ds.coverage.value = (double)rph;
    jtk_hmm_t h{};
    h.mat_mat = h.ins_mat = h.del_mat = 0.97;
    h.mat_ins = h.mat_del = h.ins_ins = h.ins_del = h.del_ins = h.del_del = 0.01;
    for (int r = 0; r < 4; r++)
        for (int q = 0; q < 4; q++) h.mat_emit[4 * r + q] = r == q ? 0.97 : 0.01;
    for (int i = 0; i < 20; i++) h.ins_emit[i] = 0.25;
    ds.model_param.forward = ds.model_param.reverse = h;
    const int n = 2 * rph;
    ds.encoded_reads.resize(n);  // "read" r carries the r-th node of every chunk
    for (int c = 0; c < n_chunks; c++) {
        jtk_synth_cfg_t cfg{};
        cfg.seed = 20260101ull + c;
        cfg.tmpl_len = tmpl_len;
        cfg.n_haps = 2;
        cfg.reads_per_hap = rph;
        cfg.min_variants = 1;
        cfg.divergence = 5e-4;
        cfg.err_sub = cfg.err_ins = cfg.err_del = 0.01;
        cfg.tmpl_err = 1e-3;
        const size_t cap = tmpl_len * 3 / 2 + 256;
        std::vector<uint8_t> tmpl(cap), reads(cap * n), ops(2 * cap * n), strand(n);
        std::vector<uint64_t> read_off(n + 1), ops_off(n + 1);
        std::vector<uint32_t> truth(n);
        uint64_t tl = 0;
        if (jtk_synth_pileup(&cfg, tmpl.data(), cap, &tl, reads.data(), reads.size(), read_off.data(), ops.data(),
                             ops.size(), ops_off.data(), strand.data(), truth.data()) != 0)
            return 4;
        jtk::Chunk chunk;
        chunk.id = c;
        chunk.seq.assign((const char *)tmpl.data(), tl);
        chunk.copy_num = 2;
        chunk.cluster_num = 2;
        ds.selected_chunks.push_back(chunk);
        for (int r = 0; r < n; r++) {
            jtk::Node node;
            node.chunk = c;
            node.seq.assign((const char *)reads.data() + read_off[r], read_off[r + 1] - read_off[r]);
            node.is_forward = strand[r] != 0;
            node.cigar = jtk::kiley_op_to_ops(ops.data() + ops_off[r], ops_off[r + 1] - ops_off[r]);
            node.posterior.assign(2, std::log(0.5));  // Node::new (definitions/src/lib.rs:713-733)
            ds.encoded_reads[r].nodes.push_back(node);
        }
    }
    jtk::LocalClusteringOptions opt;
    opt.gains = device_gains ? nullptr : &gains;
    opt.refit_model = argc > 6 && atoi(argv[6]) != 0;  // update_models_on_both_strands (mod.rs:58)
    if (getenv("JTK_HOST_MIRROR_RECORD")) opt.record = &std::cerr;  // the reference's RECORD lines (mod.rs:121) on stderr
    if (getenv("JTK_HOST_MIRROR_TRACE")) opt.trace = &std::cerr;    // and its trace! rows (TOTAL / CAND / PICK / DUMP / RANGE / LK / COUNTS)
    try {
        if (n_selected >= n_chunks) {
            jtk::local_clustering(ds, opt);
        } else {
            std::unordered_set<uint64_t> sel;
            for (int c = 0; c < n_selected; c++) sel.insert(c);
            jtk::local_clustering_selected(ds, sel, opt);
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    for (const auto &c : ds.selected_chunks)
        printf("CHUNK\t%llu\t%zu\t%.17g\t%s\n", (unsigned long long)c.id, c.cluster_num, c.score, c.seq.c_str());
    for (size_t r = 0; r < ds.encoded_reads.size(); r++)
        for (const auto &nd : ds.encoded_reads[r].nodes) {
            printf("NODE\t%zu\t%llu\t%llu\t%s", r, (unsigned long long)nd.chunk, (unsigned long long)nd.cluster,
                   cigar_string(nd.cigar).c_str());
            for (double p : nd.posterior) printf("\t%.17g", p);
            printf("\n");
        }
    return 0;
}
