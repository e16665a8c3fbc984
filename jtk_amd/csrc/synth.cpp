// synth.cpp -- host-only synthetic pile-up generator for bench.py / tests.
//
// Stands in for what the upstream JTK stages hand to local clustering: a chunk sequence (here: haplotype 0
// with a little residual error, like a once-polished chunk of determine_chunks.rs:79-140), reads drawn
// from k haplotypes with an ONT/HiFi-like error model (definitions/src/lib.rs:920-951), and per-read
// alignment ops (here: banded unit-cost global alignment, standing in for minimap2 cigar + edlib ends,
// encode/mod.rs:181-246).  The usage pattern follows sandbox/src/bin/benchmark_clustering.rs:55-100 and
// gen_sim_genome.rs:24-28 (variant rate 5e-4 split 1/3 sub/ins/del).  Not part of the hot path.
#include <cstdint>
#include <cstring>
#include <vector>

#include "jtk_lc.h"
#include "jtk_synth.h"

namespace {

struct Xoshiro {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        x += 0x9e3779b97f4a7c15ULL;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    explicit Xoshiro(uint64_t seed) {
        for (auto &v : s) v = splitmix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * 0x1p-53; }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

const char BASES[] = "ACGT";

// copy `src` introducing substitutions / insertions / deletions at the given per-base rates
void mutate(Xoshiro &rng, const std::vector<uint8_t> &src, double sub, double ins, double del,
            std::vector<uint8_t> &out, int *n_events) {
    out.clear();
    int ev = 0;
    for (uint8_t b : src) {
        double u = rng.uniform();
        if (u < sub) {
            uint8_t nb;
            do nb = (uint8_t)BASES[rng.below(4)];
            while (nb == b);
            out.push_back(nb);
            ev++;
        } else if (u < sub + ins) {
            out.push_back((uint8_t)BASES[rng.below(4)]);
            out.push_back(b);
            ev++;
        } else if (u < sub + ins + del) {
            ev++;
        } else {
            out.push_back(b);
        }
    }
    if (n_events) *n_events = ev;
}

// banded global unit-cost alignment of read to template; ties: diagonal, Del, Ins. Returns ops.
bool banded_ops(const uint8_t *t, size_t tl, const uint8_t *r, size_t rl, int band, std::vector<uint8_t> &ops) {
    // cell (i,j) kept iff |j - i*rl/tl| <= band ; store direction per cell
    const int W = 2 * band + 1;
    const uint32_t INF = 1u << 30;
    std::vector<uint32_t> prev(W + 2, INF), cur(W + 2, INF);
    std::vector<uint8_t> dir((tl + 1) * (size_t)W, 0);
    auto center = [&](size_t i) { return (int64_t)((i * rl) / (tl ? tl : 1)); };
    // row 0
    {
        int64_t c = center(0);
        for (int w = 0; w < W; w++) {
            int64_t j = c - band + w;
            if (j >= 0 && j <= (int64_t)rl) {
                prev[w] = (uint32_t)j;
                dir[w] = 2;
            }
        }
    }
    for (size_t i = 1; i <= tl; i++) {
        int64_t c = center(i), cp = center(i - 1);
        std::fill(cur.begin(), cur.end(), INF);
        for (int w = 0; w < W; w++) {
            int64_t j = c - band + w;
            if (j < 0 || j > (int64_t)rl) continue;
            uint32_t best = INF;
            uint8_t d = 0;
            int64_t wd = j - 1 - (cp - band);  // (i-1, j-1)
            if (j >= 1 && wd >= 0 && wd < W && prev[wd] < INF) {
                uint32_t v = prev[wd] + (t[i - 1] != r[j - 1]);
                if (v < best) {
                    best = v;
                    d = 0;
                }
            }
            int64_t wu = j - (cp - band);  // (i-1, j): Del
            if (wu >= 0 && wu < W && prev[wu] < INF) {
                uint32_t v = prev[wu] + 1;
                if (v < best) {
                    best = v;
                    d = 1;
                }
            }
            if (w >= 1 && cur[w - 1] < INF) {  // (i, j-1): Ins
                uint32_t v = cur[w - 1] + 1;
                if (v < best) {
                    best = v;
                    d = 2;
                }
            }
            cur[w] = best;
            dir[i * (size_t)W + w] = d;
        }
        std::swap(prev, cur);
    }
    int64_t i = (int64_t)tl, j = (int64_t)rl;
    {
        int64_t w = j - (center(tl) - band);
        if (w < 0 || w >= W || prev[w] >= INF) return false;
    }
    ops.clear();
    while (i > 0 || j > 0) {
        int64_t w = j - (center((size_t)i) - band);
        if (w < 0 || w >= W) return false;
        uint8_t d = dir[(size_t)i * W + (size_t)w];
        if (i == 0) d = 2;
        if (d == 0) {
            ops.push_back(t[i - 1] == r[j - 1] ? JTK_OP_MATCH : JTK_OP_MISMATCH);
            i--;
            j--;
        } else if (d == 1) {
            ops.push_back(JTK_OP_DEL);
            i--;
        } else {
            ops.push_back(JTK_OP_INS);
            j--;
        }
    }
    for (size_t a = 0, b = ops.size(); a + 1 < b; a++, b--) std::swap(ops[a], ops[b - 1]);
    return true;
}

}  // namespace

extern "C" {



// Generates one pile-up.  Buffers: tmpl (cap tmpl_cap), reads (cap reads_cap), ops (cap ops_cap),
// read_off/ops_off [n+1], strand[n], truth[n] (haplotype of each read), n = n_haps*reads_per_hap.
// Returns 0, or -1 if a capacity is too small / an alignment left the band.
int jtk_synth_pileup(const jtk_synth_cfg_t *cfg, uint8_t *tmpl, uint64_t tmpl_cap, uint64_t *tmpl_len_out,
                     uint8_t *reads, uint64_t reads_cap, uint64_t *read_off, uint8_t *ops, uint64_t ops_cap,
                     uint64_t *ops_off, uint8_t *strand, uint32_t *truth) {
    Xoshiro rng(cfg->seed);
    const size_t L = cfg->tmpl_len;
    std::vector<std::vector<uint8_t>> haps(cfg->n_haps);
    haps[0].resize(L);
    for (auto &b : haps[0]) b = (uint8_t)BASES[rng.below(4)];
    for (uint32_t h = 1; h < cfg->n_haps; h++) {
        int ev = 0;
        for (int tries = 0; tries < 1000; tries++) {
            mutate(rng, haps[0], cfg->divergence / 3, cfg->divergence / 3, cfg->divergence / 3, haps[h], &ev);
            if ((uint32_t)ev >= cfg->min_variants) break;
        }
    }
    std::vector<uint8_t> t;
    mutate(rng, haps[0], cfg->tmpl_err / 3, cfg->tmpl_err / 3, cfg->tmpl_err / 3, t, nullptr);
    if (t.size() > tmpl_cap) return -1;
    memcpy(tmpl, t.data(), t.size());
    *tmpl_len_out = t.size();
    uint64_t ro = 0, oo = 0;
    size_t idx = 0;
    std::vector<uint8_t> rd, op;
    read_off[0] = 0;
    ops_off[0] = 0;
    // haplotype of each read: reads_per_hap of each, in a random order (as reads arrive in a real run)
    std::vector<uint32_t> hap_of;
    for (uint32_t h = 0; h < cfg->n_haps; h++)
        for (uint32_t k = 0; k < cfg->reads_per_hap; k++) hap_of.push_back(h);
    for (size_t a = hap_of.size(); a > 1; a--) std::swap(hap_of[a - 1], hap_of[rng.below((uint32_t)a)]);
    for (; idx < hap_of.size(); idx++) {
        const uint32_t h = hap_of[idx];
        mutate(rng, haps[h], cfg->err_sub, cfg->err_ins, cfg->err_del, rd, nullptr);
        strand[idx] = (uint8_t)(rng.next() >> 63);
        truth[idx] = h;
        if (!banded_ops(t.data(), t.size(), rd.data(), rd.size(), 200, op)) return -1;
        if (ro + rd.size() > reads_cap || oo + op.size() > ops_cap) return -1;
        memcpy(reads + ro, rd.data(), rd.size());
        memcpy(ops + oo, op.data(), op.size());
        ro += rd.size();
        oo += op.size();
        read_off[idx + 1] = ro;
        ops_off[idx + 1] = oo;
    }
    return 0;
}

}  // extern "C"
